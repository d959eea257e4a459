// (second half of kernels_batch.h: the merged-launch kernel and the batch API; included behind the kernel bodies)
#pragma once
template <bool PAIR>
__global__ void __launch_bounds__(1024) PG_KERNEL_ATTR k_batch(BatchArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int w = blockIdx.x;
    int i = 0;
    while (i + 1 < A.n && w >= A.e[i].wg_end) ++i;            // (uniform: a handful of entries)
    const BatchEntryHdr& h = A.e[i];
    const int local = w - (i ? A.e[i - 1].wg_end : 0);
    const BX B = {local % h.gx, h.gx, local / h.gx, h.gy};
    const unsigned char* arg = A.blob + h.off;
    if constexpr (!PAIR) {
        const upk_igraph_t& G = *(const upk_igraph_t*)arg;
        const upk_rotamer_t& R = *(const upk_rotamer_t*)arg;
        switch (h.kind) {
            case BK_CHECK: d_pairlist_check(G, B, lds); break;
            case BK_BUILD_ROT: d_pairlist_build<true, UPK_IT_ROTAMER>(G, h.i0, h.i1, B, lds); break;
            case BK_BUILD_COV: d_pairlist_build<true, UPK_IT_HBOND_COVERAGE>(G, h.i0, h.i1, B, lds); break;
            case BK_BUILD_ENV: d_pairlist_build<true, UPK_IT_ENVIRONMENT>(G, h.i0, h.i1, B, lds); break;
            case BK_BUILD_HB: d_pairlist_build<true, UPK_IT_PROTEIN_HBOND>(G, h.i0, h.i1, B, lds); break;
            case BK_REFINE: d_pairlist_refine<false, unsigned short>(G, h.i0, h.i1, B, lds); break;
            case BK_REFINE_SYM: d_pairlist_refine<true, int>(G, h.i0, h.i1, B, lds); break;
            case BK_REFINE_SHORT: d_pairlist_refine_short(G, h.i0, B, lds); break;
            case BK_ORDER: d_pairlist_order(G, h.i0, B, lds); break;
            case BK_CLEAR_SLOTS: d_rotamer_clear_slots(R, B, lds); break;
            case BK_BUILD_SLOTS: d_rotamer_build_slots(R, B, lds); break;
            case BK_NBR_SLOTS: d_rotamer_nbr_slots(R, B, lds); break;
            case BK_SLOTS_BOTH: {      // numbering, then the stamping of the same systems by the same workgroup (what it reads it wrote itself)
                d_rotamer_build_slots(R, B, lds);
                __syncthreads();
                const BX B1 = {0, 1, B.by, B.gy};
                d_rotamer_nbr_slots(R, B1, lds);
            } break;
            default: break;
        }
    } else {
        const upk_igraph_t& G = *(const upk_igraph_t*)arg;
        const PairArgs& P = *(const PairArgs*)(arg + ((sizeof(upk_igraph_t) + 15) & ~(size_t)15));
        switch (h.kind) {
            case BK_ROWS_HB_FWD: d_pair_rows<UPK_IT_PROTEIN_HBOND, 3, 0, false>(G, P, B, lds); break;
            case BK_ROWS_HB_BWD: d_pair_rows<UPK_IT_PROTEIN_HBOND, 3, 2, false>(G, P, B, lds); break;
            case BK_ROWS_ENV_FWD: d_pair_rows<UPK_IT_ENVIRONMENT, 1, 0, false>(G, P, B, lds); break;
            case BK_COV_ROWS2: d_cov_rows2<2, false>(G, P, B, lds); break;
            case BK_COV_ROWS2_POLY: d_cov_rows2<2, true>(G, P, B, lds); break;
            case BK_ENV_BWD: d_pair_backward<UPK_IT_ENVIRONMENT, 1, false>(G, P, B, lds); break;
            case BK_COV_BWD2: d_cov_backward2<2, false>(G, P, B, lds); break;
            case BK_COV_BWD2_POLY: d_cov_backward2<2, true>(G, P, B, lds); break;
            case BK_BWD_FINISH: d_pair_backward_finish(G, h.i0, h.d0, B, lds); break;
            default: break;
        }
    }
}

extern "C" void* upk_batch_create() { return new BatchState; }
extern "C" void upk_batch_destroy(void* b) {
    BatchState* s = (BatchState*)b;
    if (s && getenv("UPSIDE_HIP_FUSE_STATS")) fprintf(stderr, "merged launches: %ld launches for %ld kernels\n", s->n_merged, s->n_items);
    delete s;
}
extern "C" int upk_batch_begin(const upk_launch_t* L) {
    BatchState* s = batch_of(L);
    if (!s) return 0;
    UPK_FLUSH(L);
    s->open = true; s->chain = 0; s->chains.clear(); s->chain_fused.clear(); s->skip_nbr_slots = false;
    return 0;
}
extern "C" void upk_batch_chain(const upk_launch_t* L, int chain) { BatchState* s = batch_of(L); if (s) s->chain = chain; }
// run what the batch holds, stage by stage
extern "C" int upk_batch_run(const upk_launch_t* L) {
    BatchState* s = batch_of(L);
    if (!s || !s->open) return 0;
    size_t n_stage = 0;
    for (auto& c : s->chains) n_stage = c.size() > n_stage ? c.size() : n_stage;
    for (size_t st = 0; st < n_stage; ++st) {
        // the stage's items, upkeep kinds and pair kinds apart (different kernels: the upkeep bodies own static LDS), split when the
        // argument block is full
        for (int pair = 0; pair < 2; ++pair) {
            BatchArgs A; memset(&A, 0, sizeof(A));
            size_t used = 0, lds = 0; int wg = 0;
            auto launch = [&]() {
                if (!A.n) return;
                if (lds < 64) lds = 64;
                if (pair) hipLaunchKernelGGL(k_batch<true>, dim3(wg), dim3(1024), lds, ST(L), A);
                else hipLaunchKernelGGL(k_batch<false>, dim3(wg), dim3(1024), lds, ST(L), A);
                s->n_merged += 1; s->n_items += A.n;
                memset(&A, 0, sizeof(A)); used = 0; lds = 0; wg = 0;
            };
            for (auto& c : s->chains) {
                if (st >= c.size() || (int)bk_is_pair(c[st].kind) != pair) continue;
                const BatchItem& it = c[st];
                const size_t need = (it.args.size() + 15) & ~(size_t)15;
                if (A.n == BATCH_MAX_ENTRIES || used + need > BATCH_BLOB) launch();
                BatchEntryHdr& h = A.e[A.n++];
                h.kind = it.kind; h.gx = it.gx; h.gy = it.gy; h.i0 = it.i0; h.i1 = it.i1; h.d0 = it.d0; h.off = (int)used;
                memcpy(A.blob + used, it.args.data(), it.args.size()); used += need;
                wg += it.gx * it.gy; h.wg_end = wg;
                if (it.lds > lds) lds = it.lds;
            }
            launch();
        }
    }
    s->chains.clear();
    return launch_status();
}
extern "C" int upk_batch_end(const upk_launch_t* L) {
    BatchState* s = batch_of(L);
    if (!s || !s->open) return 0;
    const int r = upk_batch_run(L);
    s->open = false;
    return r;
}
