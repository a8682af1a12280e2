export TMPDIR=/tmp
mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/r6_limits scripts/probes/r6_limits.hip 2>/dev/null
/opt/rocm/bin/rocm-smi --showmaxpower --showpower --showclocks 2>&1 | grep -v "^$" | head -30 > gpurun_out/r6_smi.txt
bash scripts/smi_sample.sh "A gaussian+zeros" /tmp/r6_limits A > gpurun_out/r6_limits_A_smi.txt 2>&1
/tmp/r6_limits A > gpurun_out/r6_limits_A.txt 2>&1
/tmp/r6_limits B > gpurun_out/r6_limits_B.txt 2>&1
cat gpurun_out/r6_smi.txt gpurun_out/r6_limits_A_smi.txt gpurun_out/r6_limits_A.txt gpurun_out/r6_limits_B.txt
