for kb in 16 8; do echo KB=$kb; NND_CORR_KB=$kb timeout -k 10 100 python scripts/prof_corr_build.py 2>&1 | grep -v amdgpu.ids | head -2; done
