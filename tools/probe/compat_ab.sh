#!/bin/bash
# A/B of compat-integrator builds on ONE device in ONE call: usage bash tools/probe/compat_ab.sh libA.so libB.so ...
for round in 1 2 3; do
  for lib in "$@"; do
    CATINT_PNP_LIB=$PWD/$lib python tools/probe/compat_ab.py 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib round $round: ' + '  '.join('%s %.4g (%.12g)' % (k, v['timesteps_per_s'], v['checksum']) for k, v in d.items()))"
  done
done
