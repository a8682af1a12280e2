export TMPDIR=/tmp; mkdir -p gpurun_out/r06f
for seed in 1 2 3 4 5 6 7 8; do FAD_FUZZ_SEED=$seed timeout 300 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k wide_chain -s 2>&1 | grep -E "worst relative|passed|failed|AssertionError|^case .* rel=[0-9.]+e-0[0-6]" | cut -c1-220 | sed "s/^/seed $seed: /"; done | tee gpurun_out/r06f/soak.txt
