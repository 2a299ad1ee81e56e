"""On-disk layout of the reference's results folder (catint/catint_io.py:77-91): nine pickles named alldata, species, system,
descriptors, xmesh, tmesh, electrode_reactions, electrolyte_reactions, comsol_outputs -- so the reference's plotting tools
(tools/plotting_catint.py) read a sweep produced on the GPU unchanged.  Field contract of alldata[i]: comsol_reader.py
(SURVEY.md App. D)."""
import os
import pickle

import numpy as np

NAMES = ['alldata', 'species', 'system', 'descriptors', 'xmesh', 'tmesh', 'electrode_reactions', 'electrolyte_reactions',
         'comsol_outputs']


def _plain(obj):
    """numpy scalars/arrays -> python floats / lists where the reference stores lists (comsol_reader.py appends floats)."""
    if isinstance(obj, dict):
        return obj.__class__((k, _plain(v)) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, np.generic):
        return obj.item()
    return obj


def save_all(tp, folder, only=None):
    """catint_io.save_all: writes <folder>/<name>.pkl for the nine names (or alldata under the name `only`)."""
    os.makedirs(folder, exist_ok=True)

    def dump(obj, name):
        with open(os.path.join(folder, name + '.pkl'), 'wb') as f:
            pickle.dump(obj, f, pickle.HIGHEST_PROTOCOL)
    if only is not None:
        dump(_plain(tp.alldata), only)
        return
    dump(_plain(tp.alldata), 'alldata')
    dump(_plain(tp.species), 'species')
    dump(_plain({k: v for k, v in tp.system.items() if not callable(v)}), 'system')
    dump(_plain(tp.descriptors), 'descriptors')
    dump(np.asarray(tp.xmesh), 'xmesh')
    dump(np.asarray(getattr(tp, 'tmesh', np.arange(0, 1, 0.1))), 'tmesh')
    dump(_plain(getattr(tp, 'electrode_reactions', {})), 'electrode_reactions')
    dump(_plain(getattr(tp, 'electrolyte_reactions', getattr(tp, 'reactions', {}))), 'electrolyte_reactions')
    dump(list(getattr(tp, 'comsol_outputs', ['concentrations', 'electrostatics', 'electrode_flux'])), 'comsol_outputs')


def read_all(tp, folder, only=None):
    """catint_io.read_all (which asks for 'all_data' although save_all wrote 'alldata'; both names are accepted here)."""
    def load(name):
        with open(os.path.join(folder, name + '.pkl'), 'rb') as f:
            return pickle.load(f)
    if only is not None:
        for o in ([only] if isinstance(only, str) else only):
            setattr(tp, o, load(o))
        return tp
    tp.alldata = load('alldata') if os.path.exists(os.path.join(folder, 'alldata.pkl')) else load('all_data')
    tp.all_data = tp.alldata
    for name in ('species', 'system', 'descriptors', 'xmesh', 'tmesh', 'electrode_reactions', 'electrolyte_reactions'):
        setattr(tp, name, load(name))
    tp.xmax = max(tp.xmesh)
    tp.nx = len(tp.xmesh)
    tp.dx = tp.xmesh[1] - tp.xmesh[0]
    tp.tmax = max(tp.tmesh)
    tp.nt = len(tp.tmesh)
    tp.dt = tp.tmesh[1] - tp.tmesh[0]
    return tp
