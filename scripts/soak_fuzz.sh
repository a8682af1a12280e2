#!/bin/bash
# the Frechet and per-song fuzz tests of tests/test_gpu_fuzz.py with other seeds than the committed ones (FAD_FUZZ_SEED)
export TMPDIR=/tmp; mkdir -p gpurun_out/r06i
for seed in 101 102 103 104; do FAD_FUZZ_SEED=$seed timeout 400 python -m pytest tests/test_gpu_fuzz.py -m gpu -q -p no:cacheprovider -k "fuzz_frechet_against_oracle or per_song_scores" 2>&1 | grep -E "passed|failed|Error|assert" | cut -c1-260 | sed "s/^/seed $seed: /"; done | tee gpurun_out/r06i/soak.txt
