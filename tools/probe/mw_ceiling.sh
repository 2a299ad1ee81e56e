#!/bin/bash
# step_kernel_mw with the tridiagonal solve compiled out (-DMW_NOSOLVE: results wrong) against the shipped kernel, one device, one call:
# what the kernel's memory side -- row staging through LDS, Poisson scans, stores, ~10 barriers per species row -- allows at one
# workgroup per CU.  GPU box.  usage: bash tools/probe/mw_ceiling.sh
R=$PWD; D=/tmp/mwceil; mkdir -p $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DMW_NOSOLVE catint_amd/csrc/pnp_kernels.hip -o $D/pnp_kernels.o || exit 1
objs=$(ls catint_amd/lib/obj/*.o | grep -v pnp_kernels.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $D/pnp_kernels.o $objs -o $D/libnosolve.so || exit 1
for lib in catint_amd/lib/libcatint_pnp.so $D/libnosolve.so catint_amd/lib/libcatint_pnp.so $D/libnosolve.so; do
  echo "== $lib"
  CATINT_PNP_LIB=$lib python tools/probe/mw_probe.py 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('  B=%d N=%d nx=%d spl=%d: %.1f us/step frac %.4f' % (r['B'], r['N'], r['nx'], r['steps_per_launch'], r['us_per_step'], r['frac']))"
done
