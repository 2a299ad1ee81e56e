"""The C-ABI header is plain C (gcc -std=c99 compiles it) and the ctypes mirror in catint_amd/_capi.py has the same struct layouts:
sizes and the offset of every field, taken from the compiler, not from a hand-kept table."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAIRS = {'pnp_config': 'PnpConfig', 'pnp_newton_params': 'PnpNewtonParams', 'pnp_scf_params': 'PnpScfParams',
         'pnp_scf_state': 'PnpScfState', 'pnp_ode_params': 'PnpOdeParams'}


def header_structs():
    """{struct name: [field names]} parsed from include/catint_pnp.h (typedef struct { ... } name;)."""
    src = open(os.path.join(ROOT, 'include', 'catint_pnp.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for body, name in re.findall(r'typedef\s+struct\s*\w*\s*\{(.*?)\}\s*(\w+)\s*;', src, flags=re.S):
        fields = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            names = decl.split(None, 1)[1] if not decl.startswith('const') else decl.split(None, 2)[2]
            fields += [n.strip().lstrip('*').strip() for n in names.split(',')]
        out[name] = fields
    return out


@pytest.fixture(scope='module')
def compiler_layout(tmp_path_factory):
    structs = header_structs()
    assert set(PAIRS) <= set(structs)
    d = tmp_path_factory.mktemp('abi')
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "catint_pnp.h"', 'int main(void) {']
    for s in PAIRS:
        lines.append('  printf("%s sizeof %%zu\\n", sizeof(%s));' % (s, s))
        for f in structs[s]:
            lines.append('  printf("%s %s %%zu\\n", offsetof(%s, %s));' % (s, f, s, f))
    lines += ['  return 0;', '}']
    (d / 'abi.c').write_text('\n'.join(lines))
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'), str(d / 'abi.c'), '-o', str(d / 'abi')])
    out = subprocess.check_output([str(d / 'abi')]).decode()
    layout = {}
    for line in out.splitlines():
        s, f, v = line.split()
        layout.setdefault(s, {})[f] = int(v)
    return layout


@pytest.mark.parametrize('cname', sorted(PAIRS))
def test_ctypes_mirror_matches_the_compiler(cname, compiler_layout):
    from catint_amd import _capi
    cls = getattr(_capi, PAIRS[cname])
    want = dict(compiler_layout[cname])
    assert C.sizeof(cls) == want.pop('sizeof')
    got = {n: getattr(cls, n).offset for n, _ in cls._fields_}
    assert got == want          # same field names, same offsets
