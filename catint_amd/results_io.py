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


# -- COMSOL text tables (optional export) ---------------------------------------------------------------------
# The reference's reader (catint/comsol_reader.py:125-326) parses the plain-text tables COMSOL's data export writes: comment lines
# starting with '%' up to '% Description', then ONE header line naming the columns `<variable> (<unit>) @ <parameter>=<value>`, then
# rows `x value value ...`; '% Nodes: 2' marks a table evaluated on the two boundary points (only the row at x = 0 is read).  The
# reference asks for three tables (comsol_model.py:120-125): concentrations.txt (cp1..cpN), electrostatics.txt (phi, es.Ex),
# electrode_flux.txt (j1..jN on the boundary).  export_comsol_text writes them for one descriptor point, so the reference's own
# Reader -- and every tool behind it -- can take a GPU solve where a COMSOL run used to be.
def _comsol_table(path, x, columns, names, units, par_name, par_value, nodes, description):
    with open(path, 'w') as f:
        f.write('%% Model:              %s\n' % 'pnp_transport.mph')
        f.write('% Version:            catint_amd (COMSOL data export layout)\n')
        f.write('% Dimension:          1\n')
        f.write('%% Nodes:              %d\n' % nodes)
        f.write('%% Expressions:        %d\n' % len(names))
        f.write('%% Description:        %s\n' % description)
        head = '% x' + ' ' * 22
        for n, u in zip(names, units):
            head += ' %s (%s) @ %s=%s    ' % (n, u, par_name, repr(float(par_value)))
        f.write(head.rstrip() + '\n')
        for i in range(len(x)):
            f.write(' '.join([repr(float(x[i]))] + [repr(float(col[i])) for col in columns]) + '\n')


def export_comsol_text(tp, index, folder, par_name='flux_factor', par_value=1.0):
    """Write concentrations.txt, electrostatics.txt and electrode_flux.txt of descriptor point `index` (tp.alldata[index], filled by
    Calculator.run / fill_alldata) in the layout the reference's comsol_reader.Reader parses."""
    os.makedirs(folder, exist_ok=True)
    d = tp.alldata[index]
    names = list(tp.species.keys())
    x = np.asarray(tp.xmesh, float)
    _comsol_table(os.path.join(folder, 'concentrations.txt'), x, [np.asarray(d['species'][sp]['concentration'], float) for sp in names],
                  ['cp%d' % (k + 1) for k in range(len(names))], ['mol/m^3'] * len(names), par_name, par_value, len(x), 'Concentrations')
    _comsol_table(os.path.join(folder, 'electrostatics.txt'), x,
                  [np.asarray(d['system']['potential'], float), np.asarray(d['system']['efield'], float)], ['phi', 'es.Ex'], ['V', 'V/m'],
                  par_name, par_value, len(x), 'Potential, Field')
    flux = [float(d['species'][sp].get('electrode_flux', 0.0)) for sp in names]
    _comsol_table(os.path.join(folder, 'electrode_flux.txt'), [0.0, float(x[-1])], [[f, 0.0] for f in flux],
                  ['j%d' % (k + 1) for k in range(len(names))], ['mol/m^2/s'] * len(names), par_name, par_value, 2, 'Electrode flux')
    return folder


def read_comsol_text(path):
    """One COMSOL text table (the layout _comsol_table writes and the reference's comsol_reader.py:125-196 parses): comment lines start
    with '%', the last of them names the columns `<variable> (<unit>) @ <parameter>=<value>`; rows are `x value value ...`.
    Returns (x [n], {variable: column [n]})."""
    names, rows = [], []
    with open(path) as f:
        header = None
        for line in f:
            if line.startswith('%'):
                header = line
                continue
            vals = line.split()
            if vals:
                rows.append([float(v) for v in vals])
    if header is not None:
        import re
        names = re.findall(r'([A-Za-z_][\w.]*)\s*\([^)]*\)\s*@', header)
    a = np.array(rows, float)
    if a.ndim != 2 or a.shape[1] != len(names) + 1:
        raise ValueError('%s: %d data columns for the variables %s' % (path, a.shape[1] - 1 if a.ndim == 2 else 0, names))
    return a[:, 0], {n: a[:, j + 1] for j, n in enumerate(names)}


def read_surface_concentrations(tp, folder):
    """Surface concentrations c_k(x = 0) [N] from the concentrations.txt of a COMSOL(-layout) results folder -- what the reference's
    system['init_folder'] start-up takes from comsol_reader.Reader.read_all (calculator.py:303-309)."""
    x, cols = read_comsol_text(os.path.join(folder, 'concentrations.txt'))
    names = list(tp.species.keys())
    return np.array([cols['cp%d' % (k + 1)][0] for k in range(len(names))])
