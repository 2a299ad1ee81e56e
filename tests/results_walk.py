"""The accesses the reference's plotting tool makes on a results folder (tools/plotting_catint.py:274-534), as a walker that records
what it finds: {access path: [type name, length or None]}.  Used three ways: by tests/golden/make_results_manifest.py on an object
filled by the REFERENCE's own reader (catint.catint_io.read_all with the tool's `only=[...]` list, :519), by the CPU test on
catint_amd.results_io.read_all, and by the GPU test on a real sweep.  No dependency on catint_amd or on the reference."""

SPECIES_PROFILE_PROPS = ['concentration', 'activity_coefficient']                                      # :325-343 (x = xmesh)
SPECIES_SWEEP_PROPS = ['electrode_current_density', 'electrode_flux', 'surface_concentration',         # :344-370 (x = phiM)
                       'surface_activity_coefficient']
SYSTEM_PROFILE_PROPS = ['potential', 'efield', 'charge_density', 'pH']                                 # :436-446
SYSTEM_SWEEP_PROPS = ['surface_pH', 'surface_potential', 'surface_efield', 'Stern_efield']              # :448-509


def _kind(v):
    n = None
    try:
        if not isinstance(v, (str, bytes, dict)):
            n = len(v)
    except TypeError:
        pass
    return [type(v).__name__, n]


def walk(tp):
    out = {}
    out['xmesh'] = _kind(tp.xmesh)                                                    # :274
    out["descriptors['phiM']"] = _kind(tp.descriptors['phiM'])                        # :293, :347
    for key in ('bulk_pH', 'RF', 'charging_scheme', 'Stern capacitance', 'phiPZC'):   # :177, :376, :458-469
        out["system['%s']" % key] = _kind(tp.system[key])
    n_conv = 0                                                                        # :521-527
    for i in range(len(tp.alldata)):
        key0 = [key for key in tp.alldata[i]['species']][0]
        if len(tp.alldata[i]['species'][key0]) > 0:
            n_conv += 1
    out['n_converged'] = ['int', n_conv]
    d0 = tp.alldata[0]
    for sp in tp.species:                                                             # :325, :353
        out["species['%s']['diffusion']" % sp] = _kind(tp.species[sp]['diffusion'])           # :391
        out["species['%s']['bulk_concentration']" % sp] = _kind(tp.species[sp]['bulk_concentration'])   # :402, :505
        for prop in SPECIES_PROFILE_PROPS:
            y = d0['species'][sp][prop]                                               # :336-340
            _ = [yy / 1000. for yy in y]                                              # :341 (iterable of numbers)
            out["alldata[i]['species']['%s']['%s']" % (sp, prop)] = _kind(y)
        for prop in SPECIES_SWEEP_PROPS:
            if sp not in tp.electrode_reactions and prop not in ['surface_concentration', 'surface_activity_coefficient']:
                continue                                                              # :354
            if prop in d0['species'][sp]:                                             # :369
                y = [tp.alldata[i]['species'][sp][prop] for i in range(len(tp.descriptors['phiM']))]   # :370
                _ = [float(yy) for yy in y]
                out["alldata[i]['species']['%s']['%s']" % (sp, prop)] = _kind(y[0])
            else:
                out["alldata[i]['species']['%s']['%s']" % (sp, prop)] = ['missing', None]
    for prop in SYSTEM_PROFILE_PROPS:
        y = d0['system'][prop]                                                        # :437
        _ = [float(yy) for yy in y]
        out["alldata[i]['system']['%s']" % prop] = _kind(y)
    for prop in SYSTEM_SWEEP_PROPS:
        y = [tp.alldata[i]['system'][prop] for i in range(len(tp.descriptors['phiM']))]           # :509
        _ = [float(yy) for yy in y]
        out["alldata[i]['system']['%s']" % prop] = _kind(y[0])
    # surface charge density from the Stern capacitance (:449-469), pH at a grid index (:500), concentration at x (:491)
    x_she = list(tp.descriptors['phiM'])
    _ = [tp.system['Stern capacitance'] * (x_she[i] - tp.alldata[i]['system']['surface_potential'] - tp.system['phiPZC'])
         for i in range(len(x_she))]
    out["alldata[i]['system']['pH'][inx]"] = _kind(tp.alldata[0]['system']['pH'][3])
    first = [sp for sp in tp.species][0]
    out["alldata[i]['species'][sp]['concentration'][inx]"] = _kind(tp.alldata[0]['species'][first]['concentration'][3])
    return out
