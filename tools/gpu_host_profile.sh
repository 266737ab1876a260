#!/bin/bash
# where does the HOST time of a step go?  cProfile over a few eager steps
python -m cProfile -o /tmp/prof.out bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-exact --no-optim --no-profile > /dev/null 2>&1
python - <<'PY'
import pstats
p = pstats.Stats('/tmp/prof.out')
p.sort_stats('tottime').print_stats(45)
PY
