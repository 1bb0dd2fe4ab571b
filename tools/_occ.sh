python -m pytest tests -x -q -m gpu 2>&1 | tail -2
for L in libswarmenv_w5.so libswarmenv.so; do for E in 4096 16384; do
SWARM_LIB=marl_llm_amd/lib/$L python bench.py --no-cpu-baseline --envs $E --steps 100 2>/dev/null | python tools/_fmt.py $L
done; done
