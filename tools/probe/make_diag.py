#!/usr/bin/env python3
"""dev tool: build tools/probe/libcatint_pnp_diag.so -- the library with s_memtime / s_memrealtime stamps
and a workgroup census in step_kernel_rr (diagnostic build: its fences forbid overlaps the real kernel has,
so read SHARES and placement from it, never absolute run time).  Used by stamps4.py and census4.py."""
import os
import subprocess
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'catint_amd', 'csrc')
OUT = os.path.join(ROOT, 'tools', 'probe', 'libcatint_pnp_diag.so')

STAMP = ('#define STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\\n\\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: '
         '"memory"); if (b == 0 && lane == 0 && A.dbg) A.dbg[(wave * 64 + step) * 16 + (i)] = t_; } while (0)\n'
         '#define RSTAMP(slot) do { unsigned long long rt_; asm volatile("s_memrealtime %0\\n\\ts_waitcnt lgkmcnt(0)" : '
         '"=s"(rt_) :: "memory"); A.dbg[(slot)] = rt_; } while (0)\n')


def rep(s, old, new):
    assert old in s, old[:60]
    return s.replace(old, new, 1)


def main():
    k = open(os.path.join(CSRC, 'pnp_kernels.hip')).read()
    h = open(os.path.join(CSRC, 'pnp_internal.h')).read()
    c = open(os.path.join(CSRC, 'pnp_capi.hip')).read()
    h = rep(h, '  int32_t* status;      // [B]', '  int32_t* status;      // [B]\n  unsigned long long* dbg;')
    h = h.replace('"../../include/catint_pnp.h"', '"%s"' % os.path.join(ROOT, 'include', 'catint_pnp.h'))
    k = rep(k, '#include "pnp_internal.h"', '#include "pnp_internal.h"\n' + STAMP)
    a = k.index('template <int P, int W, bool CN>\n__global__')
    e = k.index('// stand-alone Poisson (read-back of tp.potential / tp.efield)')
    body = k[a:e]
    body = rep(body, '  double lw[P + 2];', '  if (A.dbg && lane == 0 && wave == 0) { RSTAMP(4096 + b * 4 + 0); unsigned hw_ = '
               '__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); unsigned xcc_ = __builtin_amdgcn_s_getreg((20 << 0) | '
               '(0 << 6) | (3 << 11)); A.dbg[4096 + b * 4 + 2] = hw_ | ((unsigned long long)xcc_ << 32); }\n  double lw[P + 2];')
    body = rep(body, '    const bool resident = single_round && step > 0;',
               '    STAMP(0);\n    if (b == 0 && lane == 0 && A.dbg) RSTAMP((wave * 64 + step) * 16 + 8);\n'
               '    const bool resident = single_round && step > 0;')
    body = rep(body, '    double acc[P];            // this wave', '    STAMP(1);\n    double acc[P];            // this wave')
    body = rep(body, '        tridiag_wave<P, 1>(ta, tc, x, strip, XS, lane);',
               '        STAMP(2);\n        tridiag_wave<P, 1>(ta, tc, x, strip, XS, lane);\n        STAMP(3);')
    body = rep(body, '    // ---- 3. the waves add up their charge contributions', '    STAMP(4);\n    // ---- 3. the waves add up')
    body = rep(body, '    // ---- 4. charge row of the new state', '    STAMP(5);\n    // ---- 4. charge row of the new state')
    body = rep(body, '    double* tmp = lin;', '    STAMP(6);\n    double* tmp = lin;')
    body = rep(body, '  const unsigned long long nan_mask = __ballot(chk != chk);',
               '  if (A.dbg && lane == 0 && wave == 0) RSTAMP(4096 + b * 4 + 1);\n  const unsigned long long nan_mask = __ballot(chk != chk);')
    k = k[:a] + body + k[e:]
    nbytes = '(4096 + 4*65536)*8'
    c = rep(c, '  a.rates = nullptr;\n  *out = h;', '  a.rates = nullptr;\n  { unsigned long long* d_; (void)hipMalloc((void**)&d_, %s); '
            '(void)hipMemset(d_, 0, %s); a.dbg = d_; }\n  *out = h;' % (nbytes, nbytes))
    c = rep(c, 'extern "C" {\n', 'extern "C" {\nint pnp_debug_dump(pnp_handle* h, unsigned long long* out) { (void)hipStreamSynchronize(h->stream); '
            'return (int)hipMemcpy(out, h->a.dbg, %s, hipMemcpyDeviceToHost); }\n' % nbytes)
    with tempfile.TemporaryDirectory() as td:
        for name, txt in (('pnp_kernels.hip', k), ('pnp_internal.h', h), ('pnp_capi.hip', c)):
            open(os.path.join(td, name), 'w').write(txt)
        subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                        os.path.join(td, 'pnp_kernels.hip'), os.path.join(td, 'pnp_capi.hip'), '-o', OUT], check=True)
    print('built', OUT)


if __name__ == '__main__':
    main()
