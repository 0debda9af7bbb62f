"""profiles/r03_k2_pmc.json from the PMC passes over the fused forward kernel K2 in both precision
modes (tools/run_kernels.py k2 CUBE NT under `rocprofv3 --pmc ...`, summarised by
tools/pmc_summary.py):  VALU instructions per wave-step, split by issue rate.

    python tools/k2_pmc_profile.py PMC_SUMMARY.json LABEL NT OUT.json [KERNEL_PREFIX [TILES]]
(KERNEL_PREFIX: 'k_bloch_rfgr_fwd<' (default) or 'k_bloch_rfgr_bwd<' -- the fused adjoint runs persistent waves, so its
wave-steps are TILES x NT, not SQ_WAVES x NT)

Issue cost model (MI355X_MICROARCH.md: a wave64 fp32 VALU instruction occupies its SIMD-32 for 2
cycles when other waves fill the gaps; the fp64 vector rate is half the fp32 rate): fp64 FMAs and
fp32<->fp64 conversions take 4 cycles, everything else 2.  The model is checked against the run
itself: issue cycles per SIMD over the kernel's cycles (GRBM_GUI_ACTIVE / 8 XCDs) must come out
<= ~1 for a VALU-bound kernel.
"""
import json
import sys

src, label, nT, outp = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
prefix = sys.argv[5] if len(sys.argv) > 5 else 'k_bloch_rfgr_fwd<'
tiles = int(sys.argv[6]) if len(sys.argv) > 6 else None
K = json.load(open(src))[label]
modes = {}
for name, e in K.items():
    if not name.startswith(prefix):
        continue
    mode = 'precise' if 'prec_f' in name else 'fast'
    ws = (tiles if tiles else e['SQ_WAVES']) * nT             # wave-steps per launch
    per = lambda c: e.get(c, 0.0) / ws  # noqa: E731
    insts = per('SQ_INSTS_VALU')
    half = per('SQ_INSTS_VALU_FMA_F64') + per('SQ_INSTS_VALU_ADD_F64') + per('SQ_INSTS_VALU_MUL_F64') \
        + per('SQ_INSTS_VALU_CVT')
    cyc_simd = (2 * (insts - half) + 4 * half) * ws / 1024    # 1024 SIMDs
    kernel_cyc = e['GRBM_GUI_ACTIVE'] / 8
    modes[mode] = {
        'kernel': name, 'waves': e['SQ_WAVES'], 'wave_steps': ws,
        'valu_insts_per_wave_step': round(insts, 3),
        'half_rate_insts_per_wave_step': round(half, 3),
        'by_type_per_wave_step': {k_: round(per('SQ_INSTS_VALU_' + k_), 3) for k_ in
                                  ('FMA_F32', 'MUL_F32', 'ADD_F32', 'FMA_F64', 'CVT', 'INT32')},
        'issue_slots_per_wave_step': round(insts + half, 3),
        'kernel_cycles_GRBM_GUI_ACTIVE_over_8': kernel_cyc,
        'valu_issue_cycles_per_simd': cyc_simd,
        'valu_issue_frac_of_kernel_cycles': round(cyc_simd / kernel_cyc, 4),
        'raw': {k_: v for k_, v in e.items() if k_.startswith(('SQ_', 'GRBM_')) and not k_.endswith('.n')},
        'registers': {k_: e.get(k_) for k_ in ('VGPR_Count', 'SGPR_Count', 'Scratch_Size', 'LDS_Block_Size')},
    }
import os
import sys as _sys
_sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench as _bench            # one definition of the identifier: bench.source_id()
out = {
    'source_id': _bench.source_id(),       # SHA-1 over mrphy.py_amd/csrc (comments stripped): bench.py uses these counts only for the same sources
    'kernel_prefix': prefix,
    'what': 'fused forward kernel K2 (rf, gr -> Mo; no Beff in HBM), VALU instruction mix per wave-step '
            '(one wave = 64 spins, one step) from rocprofv3 PMC passes, both precision modes',
    'workload': f'{label}: nT = {nT}, fp32',
    'collected_with': 'rocprofv3 --kernel-trace --output-format csv --pmc <SQ counters> -- python3 '
                      'tools/run_kernels.py k2 CUBE NT 3   (two passes: instruction counts; per-type '
                      'counts); summarised by tools/pmc_summary.py (first dispatch of each kernel dropped)',
    'issue_cost_model': 'fp64 FMA / fp32<->fp64 conversion: 4 cycles per wave instruction (half rate); other '
                        'VALU: 2 cycles; 1024 SIMDs; kernel cycles = GRBM_GUI_ACTIVE / 8 (profiled pass)',
    'modes': modes,
}
json.dump(out, open(outp, 'w'), indent=1)
for m, e in modes.items():
    print(m, e['valu_insts_per_wave_step'], e['half_rate_insts_per_wave_step'], e['valu_issue_frac_of_kernel_cycles'])
