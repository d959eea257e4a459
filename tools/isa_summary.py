#!/usr/bin/env python3
"""Static instruction statistics of the pair kernels' inner loops (build container; needs hipcc): compiles the pair-kernel translation unit
(kernels_graph.hip) to gfx950 assembly and counts, inside the innermost loop of each named kernel, the VALU instructions, the packed ones
(v_pk_*) and the scalar fp32 arithmetic.  Writes profiles/isa_pk_share.json (read by bench.py) and a text table.

usage: python tools/isa_summary.py [round tag, default r04]"""
import collections, json, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'upside-md_amd', 'csrc')
KERNELS = {   # label -> (file, mangled-name prefix)
    'k_rotamer_grad2<false> (side-chain gradient)': ('kernels_graph.hip', '_Z15k_rotamer_grad2ILb0EE'),
    'k_cov_rows2<2,true> (coverage forward)': ('kernels_graph.hip', '_Z11k_cov_rows2ILi2ELb1EE'),
    'k_cov_backward2<2,true> (coverage backward)': ('kernels_graph.hip', '_Z15k_cov_backward2ILi2ELb1EE'),
    'k_rotamer_pair_energy<true,true> (side-chain energies, scalar form)': ('kernels_graph.hip', '_Z21k_rotamer_pair_energyILb1ELb1EE'),
    'k_rotamer_grad<true> (round-2 scalar gradient, kept as UPSIDE_HIP_PAIR2=0)': ('kernels_graph.hip', '_Z14k_rotamer_gradILb1EE'),
}
FP32_SCALAR = re.compile(r'^v_(add|sub|subrev|mul|fma|fmac|fmamk|fmaak|mac|mad)_f32')


def loop_stats(lines):
    depth = [i for i, l in enumerate(lines) if 'Depth=' in l]
    if not depth:
        return None
    dmax = max(int(re.search(r'Depth=(\d+)', lines[i]).group(1)) for i in depth)
    inner = [i for i in depth if 'Depth=%d' % dmax in lines[i]]
    lo, hi = inner[0], inner[-1]
    for j in range(hi, len(lines)):
        if re.search(r's_cbranch|s_branch', lines[j]):
            hi = j
            break
    cnt = collections.Counter()
    for l in lines[lo:hi + 1]:
        m = re.match(r'\s+([a-z_0-9]+)', l)
        if m:
            cnt[m.group(1)] += 1
    valu = sum(v for k, v in cnt.items() if k.startswith('v_'))
    pk = sum(v for k, v in cnt.items() if k.startswith('v_pk_'))
    sc = sum(v for k, v in cnt.items() if FP32_SCALAR.match(k))
    return dict(valu=valu, packed=pk, scalar_fp32_arith=sc, lds=sum(v for k, v in cnt.items() if k.startswith('ds_')),
                packed_share_of_valu=pk / max(valu, 1), packed_share_of_fp32_arith=pk / max(pk + sc, 1),
                fp32_ops_packed=2 * pk / max(2 * pk + sc, 1))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r04'
    asm = {}
    out, table = {}, []
    for label, (fn, sym) in KERNELS.items():
        if fn not in asm:
            s = '/tmp/_isa_%s.s' % fn
            subprocess.check_call(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I/opt/conda/include', '--cuda-device-only', '-S',
                                   os.path.join(CSRC, fn), '-o', s], stderr=subprocess.DEVNULL)
            asm[fn] = open(s).read().split('\n')
        L = asm[fn]
        i0 = next(i for i, l in enumerate(L) if l.startswith(sym) and l.rstrip().endswith(':') is False and ':' in l)
        i1 = next(i for i in range(i0, len(L)) if 's_endpgm' in L[i])
        st = loop_stats(L[i0:i1])
        out[label] = st
        table.append('%-78s VALU %4d  v_pk_* %4d (%.0f %% of VALU, %.0f %% of fp32 arithmetic instructions, %.0f %% of fp32 operations)  LDS %3d'
                     % (label, st['valu'], st['packed'], 100 * st['packed_share_of_valu'], 100 * st['packed_share_of_fp32_arith'], 100 * st['fp32_ops_packed'], st['lds']))
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'isa_pk_share.json'), 'w'), indent=1, sort_keys=True)
    txt = ('static instruction mix of the innermost loop (one pass = 4 pair evaluations per lane), hipcc -O3 --offload-arch=gfx950 -S\n'
           + '\n'.join(table) + '\n')
    open(os.path.join(ROOT, 'profiles', '%s_isa_summary.txt' % tag), 'w').write(txt)
    print(txt)


if __name__ == '__main__':
    main()
