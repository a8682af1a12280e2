export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -k "song or indiv or batched or g8 or g4 or fuzz or pipeline" 2>&1 | tail -4 | grep -v "^RCCL"
timeout 300 python scripts/probe_songs_general.py 2>&1 | grep "songs of"
timeout 300 python scripts/probe_c5f_prof.py 2>&1 | grep songs
