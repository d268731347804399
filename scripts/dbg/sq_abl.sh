for dbg in 0 1 2 4 3 6 7; do echo "dbg=$dbg (1 no FMA, 2 no soft-argmin, 4 no fetch)"; NND_SQ_DBG=$dbg timeout -k 10 100 python scripts/prof_squeezer.py 2>&1 | grep -v amdgpu.ids | head -1; done
