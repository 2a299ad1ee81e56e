R=$PWD; D=/tmp/floorab; mkdir -p $D; cp -r $R/catint_amd $R/tools $R/include $R/tests $R/oracle $D/
cd $D && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c -DPNP_NO_ROUNDING_FLOOR_EXIT catint_amd/csrc/pnp_newton.hip -o catint_amd/lib/obj/pnp_newton.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC catint_amd/lib/obj/*.o -o catint_amd/lib/libcatint_pnp.so || exit 1
echo "== without the rule (device), oracle with"; FUZZ_SEED=2 FUZZ_CASES=30 FUZZ_ONLY=26 python tests/fuzz/fuzz_newton.py 2>&1 | tail -2
cd $R; echo "== with the rule"; FUZZ_SEED=2 FUZZ_CASES=30 FUZZ_ONLY=26 python tests/fuzz/fuzz_newton.py 2>&1 | tail -2
