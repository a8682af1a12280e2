#!/bin/bash
# scripts/build_variant.sh <name> <extra hipcc flags...>: libfad_hip.so with frechet.hip rebuilt under the given flags -> scripts/probes/bin/libfad_<name>.so
# (the other objects are taken from fadtk_amd/build/ as the last `python -m fadtk_amd.build` left them)
name=$1; shift
B=fadtk_amd/build; O=scripts/probes/bin; mkdir -p $O
T=$(python -c "import torch, os; print(os.path.join(os.path.dirname(torch.__file__), 'lib'))")
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -Wno-unused-result "$@" -x hip -c fadtk_amd/csrc/frechet.hip -o $O/frechet_$name.o 2>&1 | grep -E "error" 
g++ -shared -fPIC -o $O/libfad_$name.so $B/common.o $B/host_stage.o $B/moments.o $B/gemm_f64.o $B/gemm_f32.o $B/frechet_f64.o $O/frechet_$name.o $B/frechet_songs.o $B/logmel.o $B/resample.o -L$T -lamdhip64 -ldl -Wl,-rpath,$T -Wl,-rpath,/opt/rocm/lib -Wl,--enable-new-dtags && echo "built $O/libfad_$name.so"
