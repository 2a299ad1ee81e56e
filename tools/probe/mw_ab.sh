#!/bin/bash
# A/B of step_kernel_mw builds on ONE device in ONE call (tools/probe/mw_probe.py per library)
for lib in "$@" "$@"; do
  echo "== $lib"
  CATINT_PNP_LIB=$PWD/$lib python tools/probe/mw_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  B=%d N=%d nx=%d spl=%d: %.1f us/step frac %.4f ok %d' % (r['B'], r['N'], r['nx'], r['steps_per_launch'], r['us_per_step'], r['frac'], r['lanes_ok']))"
done
