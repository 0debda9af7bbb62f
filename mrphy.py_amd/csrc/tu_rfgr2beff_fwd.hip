// tu_rfgr2beff_fwd.hip -- K0: launcher of mrphy_rfgr2beff
#include "host_common.hpp"

namespace {
#include "k_rfgr2beff.hpp"
}  // namespace

namespace mrphy_i {

template <typename T>
int run_rfgr2beff(const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc,
                  Bc df, Bc gam, const void* b1, void* beff, int64_t N, int64_t nM, int64_t nT,
                  int64_t nC, int store, hipStream_t st)
{
    BeffArgs<T> a;
    a.rf = (const T*)rf; a.rf_sn = rf_sn; a.gr = (const T*)gr; a.gr_sn = gr_sn;
    a.loc = (const T*)loc; a.df = df; a.gam = gam; a.b1 = (const T*)b1; a.beff = (T*)beff;
    a.nM = nM; a.nT = nT; a.nC = nC; a.c0 = 0; a.nCt = nC; a.acc = 0;
    if (N * nM * nT == 0) return 0;
    // Block order matters more than anything else here.  Blocks are dealt round-robin to the 8 XCDs,
    // so block b works on tile (b % 8) * per_xcd + b / 8: every XCD (and its L2) sweeps its own
    // contiguous eighth of Beff, time tiles fastest, i.e. 8 linear write streams.
    // measured (128^3 x 4096, ms; v = order*1000 + rows/8*10 + nt):
    //   order 0 (spin tile fastest, grid y = time tile): 128 rows+nt 16.1-17.2 | 64 rows 16.6
    //   order 1 (time tile fastest, no XCD split):       128 rows+nt 19.3
    //   order 2 (XCD sweep): 8 rows+nt 16.4 | 16+nt 14.56 | 16 14.77 | 32+nt 14.90 | 48+nt 14.92
    //                        64 15.09 | 64+nt 17.45 | 128 15.20 | 128+nt 17.88
    a.rows_per_block = 16;
    // store policy (k_common.hpp: store_pol; 16-B stores only), judged by the STEP K0 + K1 through the plain signatures in
    // fresh processes (tools/k0_store_ab.py, profiles/r05_k0_store_policy.json): `sc1 nt` below 8 GB of Beff -- it wins the
    // step by 2-4 % at 3.2 GB on boxes WITHOUT the memory-side-cache penalty of DESIGN.md §3 and by 5-8 % on boxes with
    // it -- plain `nt` from 8 GB up, where it wins or ties on the boxes without the penalty (12.9 GB: 0.1-1.8 %, 25.8 GB:
    // +-1.5 %) and the kept round-4 records chose it; a caller that owns the block can time both (workspace.BeffArena)
    a.nt = ((int64_t)3 * N * nM * nT * (int64_t)sizeof(T) < ((int64_t)8 << 30)) ? 2 : 1;
    if (store >= 0) a.nt = store;                    // the caller's choice (mrphy_rfgr2beff_st)
    int order = 2;
    if (k0_variant() > 0) {
        a.nt = k0_variant() % 10; a.rows_per_block = (k0_variant() % 1000 / 10) * 8;
        order = k0_variant() / 1000;
    }
    if (a.rows_per_block < 8) a.rows_per_block = 64;
    if (a.rows_per_block > K0_MAX_ROWS) a.rows_per_block = K0_MAX_ROWS;
    // The build: smallest register/LDS coil capacity that holds nC; everything that depends on it
    // -- elements per thread (hence the time-tile count gy), the cap on rows per block (the
    // kernel's LDS array of b1 rows) -- is read from K0Geom, the table the kernel itself uses.
    const int ncm = (nC == 1) ? 1 : (!b1 ? 0 : (nC <= 8 ? 8 : (nC <= 16 ? 16 : (nC <= K0_MAXC ? 32 : 0))));
    // 33..64 coils, float: wider capacities of the step-per-thread kernel (round 4)
    const int wide = (sizeof(T) == 4 && b1 && nC > K0_MAXC && nC <= 64) ? (nC <= 40 ? 40 : (nC <= 48 ? 48 : 64)) : 0;
    const bool vec = aligned_to(beff, sizeof(T));
    const int64_t L = 3 * nT;
    const dim3 block(K0_THREADS);
    // 2..32 coils with a map: the step-per-thread kernel (a thread owns whole time points)
    auto launch_steps = [&](auto mc_tag) -> int {
        constexpr int MC = decltype(mc_tag)::value;
        using G = K0StepGeom<T, MC>;
        if (k0_variant() <= 0 || a.rows_per_block > G::ROWS) a.rows_per_block = G::ROWS;
        const int64_t gy = (nT + (int64_t)K0_THREADS * G::TP - 1) / ((int64_t)K0_THREADS * G::TP);
        if (gy > 65535 || N > 65535) return MRPHY_EINVAL;
        const int64_t gx = (nM + a.rows_per_block - 1) / a.rows_per_block;
        dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)N);
        a.gy = 0; a.nblk = 0; a.per_xcd = 0;
        if (order >= 1 && gx * gy < (int64_t(1) << 31) - 8) {
            a.gy = (unsigned)gy; a.nblk = (unsigned)(gx * gy);
            grid = dim3(a.nblk, 1, (unsigned)N);
            if (order == 2) { a.per_xcd = (a.nblk + 7) / 8; grid.x = a.per_xcd * 8; }
        }
        hipLaunchKernelGGL((k_rfgr2beff_steps<T, MC>), grid, block, 0, st, a);
        return launch_status();
    };
    // exact coil counts 4 / 8 / 12 / 16 with a map: b1 rows as scalar operands of packed FMAs
    auto launch_pk = [&](auto nc_tag) -> int {
        constexpr int NC = decltype(nc_tag)::value;
        a.rows_per_block = 128;
        const int64_t gy = (nT + (int64_t)K0_THREADS * 2 - 1) / ((int64_t)K0_THREADS * 2);
        if (gy > 65535 || N > 65535) return MRPHY_EINVAL;
        const int64_t gx = (nM + a.rows_per_block - 1) / a.rows_per_block;
        dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)N);
        a.gy = 0; a.nblk = 0; a.per_xcd = 0;
        if (gx * gy < (int64_t(1) << 31) - 8) {
            a.gy = (unsigned)gy; a.nblk = (unsigned)(gx * gy);
            a.per_xcd = (a.nblk + 7) / 8;
            grid = dim3(a.per_xcd * 8, 1, (unsigned)N);
        }
        hipLaunchKernelGGL((k_rfgr2beff_pk<T, NC>), grid, block, 0, st, a);
        return launch_status();
    };
    // (measured, 64^3 x 1024, ms, steps kernel | this one: 4 coils 0.80 | 0.68, 8 coils 0.85 | 0.71, 12 coils
    // 0.92 | 0.90, 16 coils 0.99 | 0.86 -- and 24 coils 1.16 | 1.30, 32 coils 1.32 | 1.60: there the scalar
    // loads of the rows, 3.3-3.6 B/ns per CU when waves walk their own rows, are the wall; up to 16 only)
    if (vec && b1 && k0_pk()) {
        switch (nC) {
        case 4:  return launch_pk(std::integral_constant<int, 4>{});
        case 8:  return launch_pk(std::integral_constant<int, 8>{});
        case 12: return launch_pk(std::integral_constant<int, 12>{});
        case 16: return launch_pk(std::integral_constant<int, 16>{});
#ifdef MRPHY_DEV_KNOBS
        case 24: if (sizeof(T) == 4) return launch_pk(std::integral_constant<int, 24>{}); break;
        case 32: if (sizeof(T) == 4) return launch_pk(std::integral_constant<int, 32>{}); break;
#endif
        default: break;
        }
    }
    // More coils than the widest capacity (64 float, 32 double): blocks of that many coils, the first launch writes Beff,
    // the following ones continue the ascending FMA chains of Bx, By from the stored values -- the same chain, the same
    // bits, one read + write pass over Beff per extra block (round 4: 65 coils used to fall onto the generic kernel,
    // 264 ms at 64^3 x 1024)
    const int capmax = sizeof(T) == 4 ? 64 : K0_MAXC;
    if (vec && b1 && nC > capmax && k0_steps()) {
        for (int64_t c0 = 0; c0 < nC; c0 += capmax) {
            const int64_t nCb = nC - c0 < capmax ? nC - c0 : capmax;
            a.c0 = c0; a.nC = nCb; a.nCt = nC; a.acc = c0 > 0;
            int rc;
            // (the last block may hold few coils: still the 32-coil build, one time point per thread -- the 8- / 16-coil
            // builds' 48- / 24-byte threads read the stored values back at a third of the rate: 65 coils 7.7 vs 5.0 ms)
            if (nCb <= 32) rc = launch_steps(std::integral_constant<int, 32>{});
            else if constexpr (sizeof(T) == 4) {
                if (nCb <= 40)      rc = launch_steps(std::integral_constant<int, 40>{});
                else if (nCb <= 48) rc = launch_steps(std::integral_constant<int, 48>{});
                else                rc = launch_steps(std::integral_constant<int, 64>{});
            } else rc = MRPHY_EINVAL;
            if (rc) return rc;
        }
        return 0;
    }
    if (vec && k0_steps()) {                         // (dev knob MRPHY_K0_STEPS=0: the element-per-thread builds)
        if (ncm == 8)  return launch_steps(std::integral_constant<int, 8>{});
        if (ncm == 16) return launch_steps(std::integral_constant<int, 16>{});
        if (ncm == 32) return launch_steps(std::integral_constant<int, 32>{});
        if constexpr (sizeof(T) == 4) {
            if (wide == 40) return launch_steps(std::integral_constant<int, 40>{});
            if (wide == 48) return launch_steps(std::integral_constant<int, 48>{});
            if (wide == 64) return launch_steps(std::integral_constant<int, 64>{});
        }
    }
    auto launch = [&](auto ncm_tag) -> int {
        constexpr int NCM = decltype(ncm_tag)::value;
        using G = K0Geom<T, NCM>;
        if (a.rows_per_block > G::ROWS) a.rows_per_block = G::ROWS;
        const int vw = vec ? G::VW : 1;
        const int64_t gy = (L + (int64_t)K0_THREADS * vw - 1) / ((int64_t)K0_THREADS * vw);
        if (gy > 65535 || N > 65535) return MRPHY_EINVAL;
        const int64_t gx = (nM + a.rows_per_block - 1) / a.rows_per_block;
        dim3 grid((unsigned)gx, (unsigned)gy, (unsigned)N);
        a.gy = 0; a.nblk = 0; a.per_xcd = 0;
        if (order >= 1 && gx * gy < (int64_t(1) << 31) - 8) {
            a.gy = (unsigned)gy; a.nblk = (unsigned)(gx * gy);
            grid = dim3(a.nblk, 1, (unsigned)N);
            if (order == 2) { a.per_xcd = (a.nblk + 7) / 8; grid.x = a.per_xcd * 8; }
        }
        if (vec) hipLaunchKernelGGL((k_rfgr2beff<T, G::VW, NCM>), grid, block, 0, st, a);
        else     hipLaunchKernelGGL((k_rfgr2beff<T, 1, NCM>), grid, block, 0, st, a);
        return launch_status();
    };
    switch (ncm) {
    case 1:  return launch(std::integral_constant<int, 1>{});
    case 8:  return launch(std::integral_constant<int, 8>{});
    case 16: return launch(std::integral_constant<int, 16>{});
    case 32: return launch(std::integral_constant<int, 32>{});
    default: return launch(std::integral_constant<int, 0>{});
    }
}

}  // namespace mrphy_i

#define MRPHY_INST(T_, CT_) template int mrphy_i::run_rfgr2beff<T_>(const void* rf, int64_t rf_sn, const void* gr, int64_t gr_sn, const void* loc, Bc df, Bc gam, const void* b1, void* beff, int64_t N, int64_t nM, int64_t nT, int64_t nC, int store, hipStream_t st);
MRPHY_FOR_DATA_TYPES(MRPHY_INST)
#undef MRPHY_INST
