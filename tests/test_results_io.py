"""On-disk layout of the reference's results folder (catint/catint_io.py:77-91, :103-134) written/read by catint_amd.results_io."""
import os
import pickle

import numpy as np

from catint_amd.results_io import NAMES, read_all, save_all
from catint_amd.transport import Transport


def test_nine_pickles_round_trip(tmp_path):
    tp = Transport(descriptors={'phiM': [-0.1, -0.2]})
    tp.tmesh = np.arange(0, 1.0, 0.25)
    for i, d in enumerate(tp.alldata):
        for sp in tp.species:
            d['species'][sp] = {'concentration': np.linspace(1, 2, tp.nx) * (i + 1), 'surface_concentration': float(i + 1)}
        d['system'] = {'potential': np.zeros(tp.nx), 'surface_pH': 7.0 + i}
    folder = str(tmp_path / 'catint_results')
    save_all(tp, folder)
    assert sorted(os.listdir(folder)) == sorted(n + '.pkl' for n in NAMES)
    raw = pickle.load(open(os.path.join(folder, 'alldata.pkl'), 'rb'))
    sp0 = list(tp.species)[0]
    assert isinstance(raw, list) and isinstance(raw[1]['species'][sp0]['concentration'], list)      # lists of floats, as the reader appends
    assert raw[1]['species'][sp0]['surface_concentration'] == 2.0 and raw[1]['system']['surface_pH'] == 8.0

    class Bare(object):
        pass
    tp2 = read_all(Bare(), folder)
    assert tp2.nx == tp.nx and np.isclose(tp2.dx, tp.dx) and tp2.nt == 4 and np.isclose(tp2.dt, 0.25)
    assert list(tp2.descriptors['phiM']) == [-0.1, -0.2] and list(tp2.species) == list(tp.species)
    assert np.allclose(tp2.alldata[0]['species'][sp0]['concentration'], np.linspace(1, 2, tp.nx))
    save_all(tp, folder, only='alldata_step3')
    read_all(tp2, folder, only='alldata_step3')
    assert len(tp2.alldata_step3) == 2
