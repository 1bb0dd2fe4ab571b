python -m pytest tests -q -m gpu 2>&1 | tail -4
for E in 1280 4096 16384; do
python bench.py --no-cpu-baseline --envs $E --steps 100 2>/dev/null | python tools/_fmt.py main
done
python bench.py --no-cpu-baseline --state scatter 2>/dev/null | python tools/_fmt.py scatter
