#!/usr/bin/env python3
"""f4: a results folder written by catint_amd.results_io.save_all, read back by the REFERENCE's own reader
(catint.catint_io.read_all with the `only=[...]` list of tools/plotting_catint.py:519) and walked through the accesses that
plotting tool makes (tests/results_walk.py).  The key / type / length manifest that the reference-side walk produces is committed
(tests/golden/results_manifest.json); tests/test_results_io.py asserts it on the CPU (same synthetic sweep through our reader) and
tests/test_gpu_calculator.py on a real GPU sweep.

Dev-only (the reference is imported in a subprocess, read-only).    Usage:  python tests/golden/make_results_manifest.py
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

READER = r'''
import sys, json
sys.path.insert(0, sys.argv[3])
from catint.catint_io import read_all          # the reference's reader
import results_walk
class Object(object):
    pass
tp = Object()
read_all(tp, sys.argv[1], only=['alldata', 'species', 'system', 'xmesh', 'descriptors', 'electrode_reactions'])   # plotting_catint.py:519
json.dump(results_walk.walk(tp), open(sys.argv[2], 'w'), indent=1, sort_keys=True)
print('ok', len(tp.alldata), 'descriptor points read by the reference reader')
'''


def main():
    from tests.test_results_io import synthetic_sweep
    from catint_amd.results_io import save_all
    tmp = tempfile.mkdtemp(prefix='catint_results_')
    try:
        tp = synthetic_sweep()
        folder = os.path.join(tmp, 'CO2R_results')
        save_all(tp, folder)
        with open(os.path.join(tmp, 'reader.py'), 'w') as f:
            f.write(READER)
        env = dict(os.environ, PYTHONPATH=REF, PYTHONDONTWRITEBYTECODE='1')
        out = os.path.join(tmp, 'manifest.json')
        r = subprocess.run([sys.executable, 'reader.py', folder, out, os.path.join(ROOT, 'tests')], cwd=tmp, env=env, capture_output=True, text=True)
        print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-3000:])
        if r.returncode != 0:
            print(r.stderr[-3000:])
            raise SystemExit('reference reader failed')
        shutil.copy(out, os.path.join(HERE, 'results_manifest.json'))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == '__main__':
    main()
