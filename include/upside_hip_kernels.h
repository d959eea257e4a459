/* Thin C-ABI between the host-side C++ engine (DerivEngine / DerivComputation nodes) and the hand-written
 * HIP kernels for gfx950.  Plain pointers, ints and floats only; every pointer is a DEVICE pointer unless
 * marked host.  All launchers are asynchronous on `L->stream` and return the hipError_t of the launch
 * (0 = success).  Arrays carry a leading system dimension S = L->n_system ("[S]" below): S independent
 * replicas / ensemble members of one topology are processed by one launch (grid.y = system).
 *
 * Each launcher names the reference loop it replaces (paths relative to /root/reference/).
 */
#ifndef UPSIDE_HIP_KERNELS_H
#define UPSIDE_HIP_KERNELS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* fuse: the engine's queue of fused per-element ops (upk_fuse_create) or NULL.  Per-element launchers (marked "fusable" below)
 * append an op to it instead of launching; every other launcher runs the queue first (upk_fuse_flush), so the order of calls is
 * the order of effects.  A caller that enqueues its own work on `stream` must call upk_fuse_flush(L) first. */
typedef struct { int n_system; void* stream; void* fuse; void* batch; } upk_launch_t;
/* batch: merged launches (csrc/kernels_batch.h) or NULL.  Between upk_batch_begin and upk_batch_end the launchers of the list
 * upkeep and of the pair passes that have a batch form append to the chain named by upk_batch_chain instead of launching; launches
 * of one chain keep their order, the k-th launches of all chains run side by side as ONE launch.  The caller promises that
 * different chains do not depend on each other.  Any other launcher runs what the batch holds first. */
void* upk_batch_create(void);
void upk_batch_destroy(void* batch);
int upk_batch_begin(const upk_launch_t* L);
void upk_batch_chain(const upk_launch_t* L, int chain);
int upk_batch_run(const upk_launch_t* L);
void upk_batch_fused_submitted(const upk_launch_t* L);   /* (called by the fused-op queue) */
int upk_batch_end(const upk_launch_t* L);
void* upk_fuse_create(int n_system);
void upk_fuse_destroy(void* fuse);
int upk_fuse_flush(const upk_launch_t* L);          /* one launch for the pending ops (no-op when nothing is pending) */
int upk_fuse_pending(const upk_launch_t* L);        /* number of ops waiting */
long upk_fuse_launch_count(const upk_launch_t* L);  /* fused launches so far (diagnostics) */
int upk_fuse_table_size(const upk_launch_t* L);     /* distinct ops registered so far: constant once every op of the MD loop has been seen */
#define UPK_FLUSH(L) do { const int r_ = upk_fuse_flush(L); if (r_) return r_; } while (0)

/* A CoordNode's storage (src/deriv_engine.h:83-96): out/sens are [S][n_elem][stride] */
typedef struct { float* out; float* sens; int n_elem; int width; int stride; } upk_coord_t;

/* ---- generic ---------------------------------------------------------------------------------- */
/* out[s] (+)= sum_i in[s][i]; one workgroup per system, fixed tree order (deterministic).            */
int upk_reduce_sum(const upk_launch_t* L, const float* in, int n, float* out, int accumulate);
int upk_scale(const upk_launch_t* L, float* x, int n, float factor);   /* x[i] *= factor */
/* zero n_buf device buffers (float counts in sizes[], 16-byte aligned) in one launch:
 * the per-node "zero sensitivity" of deriv_engine.cpp:147-151 for the whole graph */
int upk_zero_many(const upk_launch_t* L, float* const* ptrs, const long* sizes, int n_buf);
/* deterministic gather of deferred derivative contributions into a node's sens:
 * sens[s][t][c] += sum_{e in csr[t]} arena[s][entry[e] + c], c < width  (replaces the scatter-adds of
 * e.g. src/bonds.cpp:315-316, src/placement.cpp:304-305, src/eig.cpp:467)                               */
int upk_gather_contrib(const upk_launch_t* L, const float* arena, long arena_stride, const int* csr_start,
                       const int* csr_entry, upk_coord_t target, int width, int comp_offset);

/* ---- Monte-Carlo pivot moves (src/monte_carlo_sampler.cpp:3-155, 255-284) -------------------------------- */
typedef struct {
    int n_loc, n_bin, n_layer;
    const int* atoms;          /* [n_loc][5] prevC, N, CA, C, nextN */
    const int* range;          /* [n_loc][2] rigid tail [first, end) rotated with the pivot */
    const int* restype;        /* [n_loc] layer of the proposal map */
    const float* pot;          /* [n_layer][n_bin*n_bin] -log proposal probability, normalised */
    const float* cdf;          /* [n_layer][n_bin*n_bin] cumulative proposal probability */
} upk_pivot_t;
/* every system: save the positions, draw a pivot (stream 2 keyed by `round`), rotate the tail; delta_lprob[s] out */
int upk_pivot_propose(const upk_launch_t* L, upk_coord_t pos, float* pos_copy, const upk_pivot_t* P, const uint32_t* seed,
                      uint64_t round, float* delta_lprob);
/* rigid-body jump of one chain segment (monte_carlo_sampler.cpp:157-251): translation by sigma_trans/sqrt(3) * N(0,1)^3
 * or rotation about the centre of mass by sigma_rot * N(0,1) around a random axis; random stream 3; delta_lprob = 0 */
typedef struct { int n_chain; const int* atom_range; /* [n_chain][2] */ const float* sigma_trans; const float* sigma_rot; } upk_jump_t;
int upk_jump_propose(const upk_launch_t* L, upk_coord_t pos, float* pos_copy, const upk_jump_t* J, const uint32_t* seed,
                     uint64_t round, float* delta_lprob);
/* Metropolis test of monte_carlo_step (second draw of the same generator); rejected systems get pos_copy back;
 * stats[s] = {n_success, n_attempt} accumulate */
int upk_mc_accept(const upk_launch_t* L, upk_coord_t pos, const float* pos_copy, const float* e_old, const float* e_new,
                  const float* delta_lprob, const float* temperature, const uint32_t* seed, uint64_t round, int stream,
                  int accept_draw, int* stats);   /* accept_draw: index of the generator's draw used for the test */

/* ---- integrator / thermostat (src/deriv_engine.cpp:11-48, src/thermostat.cpp:9-18, src/random.h) ---- */
int upk_integration_stage(const upk_launch_t* L, float* mom, upk_coord_t pos, float vel_factor, float pos_factor,
                          float max_force);
/* n_invocations[S] lives on the device, one (equal) entry per system -- all systems are thermalised at the same rounds --:
 * the op reads its system's entry and then advances it, so a captured MD graph replays correctly; mom is [S][n_atom][4] */
int upk_thermostat(const upk_launch_t* L, float* mom, int n_atom, const uint32_t* seed, unsigned long long* n_invocations,
                   const float* mom_scale, const float* noise_scale);
int upk_recenter(const upk_launch_t* L, upk_coord_t pos, int xy_only);
int upk_kinetic(const upk_launch_t* L, const float* mom, int n_atom, float* kin);

/* ---- backbone coordinate nodes ------------------------------------------------------------------ */
/* affine_alignment (src/eig.cpp:317-470): eig is [S][n_res][20] (4 eigenvalues + 4x4 eigenvectors) */
int upk_affine_fwd(const upk_launch_t* L, upk_coord_t pos, const int* atoms, const float* ref_geom, int n_res,
                   upk_coord_t out, float* eig);
/* writes per-residue 3 atoms x 3 comps into contrib[s][res*9 ...] */
int upk_affine_bwd(const upk_launch_t* L, upk_coord_t aff, const float* ref_geom, const float* eig, int n_res,
                   float* contrib, long contrib_stride);
/* rama_coord (src/bonds.cpp:205-247): jac [S][n_res][UPK_RAMA_JAC] -- the 2 x 5 x 3 derivatives of a residue in the first 30 floats
 * of a 128-byte row (written as eight 16-byte words: the buffer must be 16-byte aligned and hold UPK_RAMA_JAC floats per residue) */
#define UPK_RAMA_JAC 32
int upk_rama_fwd(const upk_launch_t* L, upk_coord_t pos, const int* atom, const int* dummy, int n_res, upk_coord_t out,
                 float* jac);
int upk_rama_bwd(const upk_launch_t* L, upk_coord_t rama, const float* jac, int n_res, float* contrib, long contrib_stride);
/* infer_H_O (src/hbond.cpp:59-119): dfd [S][n_virtual][12] */
int upk_infer_fwd(const upk_launch_t* L, upk_coord_t pos, const int* atom, const float* bond_length, int n_virtual,
                  upk_coord_t out, float* dfd);
int upk_infer_bwd(const upk_launch_t* L, upk_coord_t infer, const float* bond_length, const float* dfd, int n_virtual,
                  float* contrib, long contrib_stride);

/* ---- bonded potentials (src/bonds.cpp:297-318, 457-487, 519-545, 350-372) -------------------------- */
/* kind: 2 dist, 3 angle, 4 dihedral.  contrib: [term][kind][3]; pot_terms [S][n] (may be NULL)         */
int upk_spring(const upk_launch_t* L, int kind, upk_coord_t pos, const int* id, const float* equil, const float* k, int n,
               float* contrib, long contrib_stride, float* pot_terms);
int upk_cavity_radial(const upk_launch_t* L, upk_coord_t pos, const int* id, const float* radius, const float* k, int n,
                      float* contrib, long contrib_stride, float* pot_terms);

/* ---- placement (src/placement.cpp:264-307) ------------------------------------------------------- */
typedef struct {
    int n_elem, n_pos_dim, n_sig; int sig[3];              /* 0 scalar, 1 vector, 2 point */
    const int* affine_residue; const int* layer; const int* rama_residue;
    int is_rama; const float* fixed_data;                  /* (n_layer, n_pos_dim) */
    const float* spline_coeff; int nx, ny;                 /* (n_layer,nx,ny,n_pos_dim,16) */
} upk_placement_t;
int upk_placement_fwd(const upk_launch_t* L, const upk_placement_t* P, upk_coord_t aff, upk_coord_t rama, upk_coord_t out,
                      float* rama_deriv);
/* aff_contrib [elem][6] (com, torque); rama_contrib [elem][2] (NULL unless is_rama) */
int upk_placement_bwd(const upk_launch_t* L, const upk_placement_t* P, upk_coord_t aff, upk_coord_t out,
                      const float* rama_deriv, float* aff_contrib, long aff_stride, float* rama_contrib, long rama_stride);

/* ---- simple per-element nodes ---------------------------------------------------------------------- */
/* rama_map_pot (src/rama_map_pot.cpp:57-82): adds into rama.sens directly (residue ids are distinct) */
int upk_rama_map_pot(const upk_launch_t* L, upk_coord_t rama, const int* residue, const int* map_id, int n,
                     const float* coeff, int nx, float* pot_terms);
/* weighted_pos (src/environment.cpp:132-154) */
int upk_weighted_pos_fwd(const upk_launch_t* L, upk_coord_t pos, upk_coord_t energy, const int* index_pos,
                         const int* index_weight, upk_coord_t out);
int upk_weighted_pos_bwd(const upk_launch_t* L, upk_coord_t pos, upk_coord_t energy, const int* index_pos,
                         const int* index_weight, upk_coord_t self);
/* nonlinear_coupling (src/environment.cpp:358-369) */
int upk_nonlinear_coupling(const upk_launch_t* L, upk_coord_t input, const int* types, const float* coeff, int n_coeff,
                           float offset, float inv_dx, float* pot_terms);
/* hbond_energy (src/hbond.cpp:430-444) */
int upk_hbond_energy(const upk_launch_t* L, upk_coord_t protein_hbond, float E_protein, float* pot_terms);
/* backbone_pairs (src/backbone_steric.cpp:81-145): gather form over residue pairs; aff_contrib [res][6]; pot_terms [S][n_res]
 * (each pair counted once).  cache (may be NULL: every residue scans all others each step, what the reference does): per-row lists
 * of the residues within dist_cutoff + skin of the row's REFERENCE centre, rebuilt by the kernel when the two largest centre
 * displacements add up to the skin; used by the multi-workgroup launch (n_res > 128).  The caller flips `parity` on every call. */
typedef struct {
    int* list;            /* [S][n_res][cap] */
    int* cnt;             /* [S][n_res] */
    float *ref0, *ref1;   /* [S][n_res][4] reference centres, double buffered (initialised far away: the first call builds) */
    int cap, parity; float skin;
    int* error_flag;      /* set to 1 when a row outgrows cap */
} upk_backbone_list_t;
int upk_backbone_pairs(const upk_launch_t* L, upk_coord_t aff, const int* residue, const int* id, const int* n_atom,
                       const float* ref_pos, int n_res, float dist_cutoff, float* aff_contrib, long aff_stride,
                       float* pot_terms, const upk_backbone_list_t* cache);

/* ---- interaction graph (src/interaction_graph.h) ---------------------------------------------------- */
enum { UPK_IT_ROTAMER = 0, UPK_IT_HBOND_COVERAGE = 1, UPK_IT_ENVIRONMENT = 2, UPK_IT_PROTEIN_HBOND = 3,
       UPK_IT_RADIAL = 4 /* symmetric, sidechain_radial.cpp:16-79 */, UPK_IT_HBOND_SC_RADIAL = 5 /* the same functor between two nodes */ };

typedef struct {
    int itype, symmetric;
    int n1, n2;                          /* elements on each side */
    int dim1, dim2;                      /* components used (6/6, 7/6, 6/4) */
    upk_coord_t node1, node2;            /* source CoordNodes */
    const int *loc1, *loc2, *type1, *type2, *id1, *id2;
    const float* param; int n_type1, n_type2, n_param;
    int n_knot, n_knot_angular; float inv_dx, inv_dtheta;
    float cutoff, cache_cutoff;          /* cache_cutoff = cutoff + skin (interaction_graph.h:97,395-396) */
    /* cached Verlet lists: for side-1 rows nbr1[s][i][k] (neighbours on side 2, ascending) and count;
       for asymmetric graphs the transposed list nbr2 as well; symmetric graphs keep the full list in nbr1. */
    int cap1, cap2;
    int *nbr1, *cnt1, *nbr2, *cnt2;
    float *cache_pos1, *cache_pos2;      /* [S][n][4] positions the lists were built from; 4th word = element id (raw bits) */
    int* rebuild_flag;                   /* [S] system moved further than the skin allows */
    int* flagged; int parity, flag_stride; /* [2][flag_stride]: compact list of the systems flagged this step: entry 0 of
                                          * half `parity` is the count, entries 1.. the system ids; the other half is
                                          * reset for the next step.  Rebuild kernels loop over this list with a small
                                          * grid.y instead of launching (and retiring) workgroups for every system. */
    int* error_flag;                     /* [1] set to non-zero on capacity overflow */
    /* optional hook used by the rotamer node: while a symmetric list is rebuilt, mark_table[s][node(i)][node(j)]
       (mark_n rows of mark_ld bytes, cleared by the rebuild test that flags the system) is set to 1 for every cached pair */
    unsigned char* mark_table; const int* mark_node; int mark_n, mark_stride;   /* mark_stride: bytes per system (multiple of 16) */
    int mark_ld;                         /* bytes per row: mark_n rounded up to 64 (a row is whole 64-bit words of the packed bit matrix) */
    int mark_start3, mark_start6;        /* node of a bead id (rotamer.cpp:812-816): (id >> 8) + {0, mark_start3, mark_start6} by its state count */
    /* This step's in-range pairs ("hit lists", the refine of interaction_graph.h:201-257 done ONCE per step and side by
       upk_pairlist_refine): for every row of side 1 (hit1) / side 2 (hit2) the cached neighbours with d2 < cutoff2, in list
       order; hit[s][row][k] holds the list word unchanged, hcnt[s][row] their number (capacity per row = cap1 / cap2).
       Symmetric graphs: hlo1[s][row] = hits whose partner index is BELOW the row (lists ascend, so the partners above the
       row -- each pair once, i1 < i2 -- are the tail [hlo1, hcnt1) of the hit list).
       ord1 / ord2 [s][row]: rows sorted by descending hit count (upk_pairlist_order), ord1u by descending count of partners
       above the row: the pair passes hand 8 consecutive rows of that order to the 8 lane groups of a wavefront, so the
       groups of a wavefront run the same number of trips.
       The rotamer graph's list words carry the residue-pair slot above bit UPK_ROT_J_BITS. */
    int *hit1, *hit2, *hcnt1, *hcnt2, *hlo1;
    unsigned short *ord1, *ord2, *ord1u;
    float *cur_pos1, *cur_pos2;          /* [S][n][4] this step's positions (x, y, z, -), written by upk_pairlist_check */
    unsigned long long* gacc;            /* [S][n_other][8] exact fixed-point (x 2^32) gradient accumulators of upk_igraph_backward when several
                                            workgroups serve one system (small batches); NULL: one workgroup per system; zero between evaluations */
    int nbr_j_bits;                      /* 0: a list word is the element index; else the index is its low nbr_j_bits bits */
    /* quadspline graphs (hbond_coverage): the same table as `param`, every knot interval of every spline expanded on the host
       into its cubic's 4 monomial coefficients: [n_type1][n_type2][n_poly], n_poly = 8 (n_knot_angular - 3) + 8 (n_knot - 1),
       rows 16-byte aligned (layout: igraph_device.h, quadspline_pair<.., POLY>).  The LDS-staged pair passes read this one;
       NULL: they read `param`. */
    const float* param_poly; int n_poly;
    int sens_overlap;                    /* node1 and node2 are one node AND some element is listed on both sides (set by the host) */
    int word16;                          /* 1: the words of nbr1/nbr2/hit1/hit2 are 16 bits wide (bare element indices; every graph but the rotamer's:
                                            n1, n2 <= 65534), rows cap1 / cap2 such words apart in the first half of the arrays; 0: 32-bit words */
} upk_igraph_t;
#define UPK_ROT_J_BITS 13                /* rotamer list word = bead | slot << 13: <= 8191 beads (bead index n1 is the refine kernel's sentinel), < 2^19 - 1 slots */
#define UPK_ROT_SLOT_NONE 0x7FFFF        /* slot field of a cached bead pair whose residue pair got no slot (capacity overflow) */

/* K1: flag[s] |= any element moved more than (cache_cutoff-cutoff)/2 since the last build
 * (interaction_graph.h:57-90) */
int upk_pairlist_check(const upk_launch_t* L, const upk_igraph_t* G);
/* K2: where flag[s] is set rebuild the row lists with d < cache_cutoff and acceptable_id_pair
 * (interaction_graph.h:116-158); clears the flag */
int upk_pairlist_build(const upk_launch_t* L, const upk_igraph_t* G);
/* ... of the sides in `sides` only (bit 1: side-1 rows, bit 2: side-2 rows): a graph whose pair passes all gather over the rows of
 * one side never reads the other side's lists on the MD path (they can be built later from the same cache_pos) */
int upk_pairlist_build_sides(const upk_launch_t* L, const upk_igraph_t* G, int sides);
/* 1 if upk_igraph_rows and upk_igraph_backward over the rows of `row_side` take the LDS-staged (hit-list) path for this graph,
 * 0 if they fall back to the list-walking kernels, which read the cached lists of BOTH sides */
int upk_igraph_passes_staged(const upk_launch_t* L, const upk_igraph_t* G, int row_side);
/* K2b: this step's in-range pairs of every system for the rows of `side` (hit lists above), from the cached lists and
 * cur_pos; then the rows of that side sorted by descending hit count (ord1 / ord2, and ord1u for symmetric graphs) */
int upk_pairlist_refine(const upk_launch_t* L, const upk_igraph_t* G, int side);
int upk_pairlist_order(const upk_launch_t* L, const upk_igraph_t* G, int side);
/* Pair passes over the hit lists (LDS-staged; fall back to the list-walking forms below for systems too large for LDS).
 * upk_igraph_rows: for every row of `side`
 *   mode 0: out[s][(out_row0 + row)*out_stride + out_comp] = sum over the row's in-range partners of the pair value
 *           (hbond.cpp:387-389, environment.cpp:85-91, hbond.cpp:316-319)
 *   mode 1: the same, and own_grad[s][row][8] = the UNWEIGHTED sum of d(value)/d(row element): when the pair sensitivity
 *           is the row element's own (coverage nodes), that side's backward pass is upk_igraph_apply_own_grad
 *   mode 2: sens(pair) * d(value)/d(row element) summed over the partners and added to the source node's sens at
 *           loc[row] (interaction_graph.h:525-555 as a per-row gather; no atomics, fixed summation order).
 *           Pair sensitivity: sens_mode 1: sens1[s][i1*sens_stride]; 2: sens2[s][i2*sens_stride]; 3: sens1[i1]+sens2[i2].
 * side = 3: the rows of side 1, then the rows of side 2, in one launch (protein_hbond; mode 0 writes side 2 at out_row0_2) */
int upk_igraph_rows(const upk_launch_t* L, const upk_igraph_t* G, int side, int mode, float* out, long out_sys_stride, int out_stride,
                    int out_comp, int out_row0, int out_row0_2, float* own_grad, int sens_mode, const float* sens1, const float* sens2,
                    long sens_sys_stride, int sens_stride);
int upk_igraph_apply_own_grad(const upk_launch_t* L, const upk_igraph_t* G, int side, const float* own_grad,
                              const float* sens, long sens_sys_stride, int sens_stride);
/* Backward over the hit lists of side `row_side`, ONE visit per pair: sens(pair) * d(value)/d(element) is added to the source
 * nodes' sens of BOTH elements (interaction_graph.h:525-555).  The row element's share accumulates in registers, the other
 * element's through 64-bit integer LDS atomics as exact fixed point, so results do not depend on the order of the pairs. */
int upk_igraph_backward(const upk_launch_t* L, const upk_igraph_t* G, int row_side, int sens_mode, const float* sens1,
                        const float* sens2, long sens_sys_stride, int sens_stride);
/* List-walking forms (no hit lists, no LDS staging; any system size): the radial potentials and the fallback of the two
 * launchers above.  Row sums of the pair value over one side's cached lists ... */
int upk_igraph_rowsum(const upk_launch_t* L, const upk_igraph_t* G, int side, float* out, long out_sys_stride,
                      int out_stride, int out_comp, int out_row0, float* own_grad /* may be NULL; as mode 1 above */);
/* ... and the per-row gather of sens(pair) * d(value)/d(row coords), added to the source node's sens at loc[row].
 * sens_mode 0: every pair has sensitivity 1 (potentials summed over edges, sidechain_radial.cpp:94-96) */
int upk_igraph_grad(const upk_launch_t* L, const upk_igraph_t* G, int side, int sens_mode, const float* sens1,
                    const float* sens2, long sens_sys_stride, int sens_stride);
/* parity/diagnostic: flags[s][i][k] = 1 where cached neighbour k of row i is in range this step */
int upk_igraph_inrange(const upk_launch_t* L, const upk_igraph_t* G, unsigned char* flags);

/* ---- rotamer (src/rotamer.cpp) ------------------------------------------------------------------- */
/* grid.y of the rebuild kernels: enough slices that the systems flagged in one step (about one in five with the
 * short list margin) are rebuilt side by side, few enough that a quiet step retires almost no idle workgroups
 * (measured at 1024 systems, same box: S/8 91.8 k, S/4 92.8 k, S/2 92.5 k system-steps/s) */
#define UPK_FLAG_DIV_DEFAULT 4
#ifndef UPK_FLAG_DIV
#define UPK_FLAG_DIV UPK_FLAG_DIV_DEFAULT
#endif
#define UPK_FLAG_GRID(S) ((S) < 16 ? (S) : ((S) / UPK_FLAG_DIV > 16 ? (S) / UPK_FLAG_DIV : 16))
#define UPK_FLAG_LIST(G) ((G).flagged + (size_t)(G).parity * (G).flag_stride)

typedef struct {
    upk_igraph_t G;                      /* symmetric bead graph */
    int n_node, n_node1, n_node3;        /* global node ids: class 1, then 3, then 6 */
    const int* node_nrot;                /* [n_node] */
    const int *bead_node, *bead_rot;     /* [n_bead] */
    const int *bead_orig;                /* [n_bead] the bead's index in the configuration's own order (NULL: unchanged); the parameter
                                            derivative orients a pair (i1 < i2) by it, as the reference's edge list does */
    const int *bead_meta;                /* [n_bead] type | rot<<8 | n_rot<<12 (staged into the LDS bead rows) */
    const float* param_tri;              /* upper triangle of the (symmetric) pair table: row tri(lo, hi) = interaction_param[lo][hi], lo <= hi;
                                            (hi, lo) reads the same row with its two angular blocks exchanged (is_compatible, bead_interaction.h:209-218) */
    const float* param_tri_poly; int n_poly;   /* the same triangle as per-interval polynomials (upk_igraph_t::param_poly layout, n_poly floats per row); the
                                            energy pass stages it when it fits LDS next to the beads (NULL / does not fit: param_tri) */
    float* bead_pack;                    /* [S][n_bead][8] packed bead rows for systems whose beads do not fit LDS (else NULL) */
    unsigned long long* grad_acc;        /* [S][n_bead][6] exact fixed-point (x 2^32) gradient accumulators of the gradient pass when a system is
                                            served by several workgroups or does not fit LDS; zero between evaluations */
    int one_bead_per_state;              /* every (residue, rotamer state) owns exactly one bead: each pair-matrix entry has a single writer */
    int node_prob_in_solve;              /* the one-workgroup solve computes the node probabilities itself (upk_rotamer_node_prob is then a no-op): small batches */
    int p_prob;                          /* the pair-energy kernel stores exp(-E) (resting value 1) instead of E (resting value 0): the
                                            one-workgroup solve then has no exp pass.  Needs one_bead_per_state and bp_C <= 1 */
    const int *node_bead_start, *node_bead_list;   /* CSR (node*6+rot) -> beads of that rotamer state */
    int n_prob; const float* const* prob_out; float* const* prob_sens; const int* prob_stride;   /* 1-body parents (device arrays of device ptrs) */
    const long* prob_sys_stride;
    /* per system state */
    float *node_prob, *node_off, *nb_cur;               /* [S][n_node][6], off [S][n_node] */
    int slot_cap, adj_cap;
    int *n_slot, *slot_a, *slot_b, *slot_of, *slot_active;   /* [S], [S][cap], [S][cap], [S][n_node^2], [S][cap] */
    unsigned char* mark;                 /* [S][n_node][G.mark_ld] residue pairs owning a cached bead pair (= G.mark_table) */
    int *adj_cnt, *adj_slot;             /* [S][n_node], [S][n_node][adj_cap] */
    int *bp_start, *slot_off;            /* [S][n_node+1] inbox CSR of BP messages, [S][cap][2] inbox offsets of a slot */
    int *row_start, *slot_row;           /* the same inbox counted in ROWS (one per message): [S][n_node+2] first row of a node (entry n_node: end;
                                            entry n_node+1: R6, the first row of the 6-state block = rows of the 3-state nodes rounded up to 32),
                                            [S][cap][2] rows of a slot's two messages.  The large-batch solve lays the messages of the slots ACTIVE
                                            in this evaluation out densely (3 / 6 floats per row) so that they fit the LDS; NULL: not kept */
    int *class_start;                    /* [S][6] slot ranges by class: 3x3, 3x6, 6x6, 1x1, 1xN */
    int *slot_active_last;               /* [S][cap] activity flags of the last solve (diagnostics) */
    float *P, *msg_cur, *marg;           /* P, marg: SoA [S][36][cap]; msg_cur: inbox [S][cap][16] floats at most: message rows grouped by
                                          * receiving node, 4 floats per message to a 3-state node, 8 to a 6-state node */
    float damping, tol; int max_iter, chunk;
    int* iters;                          /* [S] sweeps of the last solve */
    int* n_bad;                          /* [S] solves that ran into max_iter (rotamer.cpp:784-785), counted on the device */
    int* bp_rec;                         /* [S][slot_cap][4] scratch of the one-workgroup solve: per class, the slots active this step
                                            packed as {offset a, offset b, node a | node b << 16, slot} */
    int* bp_layout;                      /* [S][7 n_node + 8] the dense inbox layout of the last solve when k_rotamer_bp_layout computes it
                                            in front of the one-workgroup solve: first float of every node [n_node + 1], active slots per class [3],
                                            inbox floats, floats of the rows to 3-state nodes, row width, pad to n_node + 8, then the node
                                            probabilities with the 1-state partners folded in [n_node][6] (raw float bits); NULL: the solve lays
                                            its inbox out and folds itself (small batches) */
    long long* bp_trace;                 /* [S][32] 100 MHz phase clocks of the last solve ([16..24): sub-phase stamps), or NULL (diagnostics) */
    float* energy;                       /* [S] Bethe free energy (only when want_energy) */
    /* cluster solve (bp_C > 1 workgroups per system, pair matrices resident in LDS) */
    int bp_C, bp_resident;               /* workgroups per system; 1 = matrices resident in LDS (512 lanes), 0 = split solve over global memory */
    int *bp_bar, *bp_fallback;           /* [S] barrier counter; [S] 1 = system did not fit or a cluster barrier gave up: solved by the one-workgroup kernel */
    int bp_test_abort;                   /* tests (UPSIDE_HIP_BP_CLUSTER_TEST_ABORT): the last workgroup of every cluster leaves at once, barriers give up early */
    float *bp_nbx, *bp_dev, *bp_en_part; /* [S][2][n_node][8] exchanged node beliefs, [S][2][16] deviations, [S][16] energy partial sums */
} upk_rotamer_t;
#define UPK_BP_LAYOUT_PER_NODE 7         /* ints of upk_rotamer_t::bp_layout per system: UPK_BP_LAYOUT_PER_NODE * n_node + UPK_BP_LAYOUT_EXTRA */
#define UPK_BP_LAYOUT_EXTRA 8

/* pair-list rebuild of flagged systems (replaces EdgeLocator, rotamer.cpp:134-206): clear the node x node table,
 * [upk_pairlist_build marks it through G.mark_table], number the slots by class + adjacency + inbox layout, and
 * record the slot of every cached bead pair */
int upk_rotamer_clear_slots(const upk_launch_t* L, const upk_rotamer_t* R);
int upk_rotamer_build_slots(const upk_launch_t* L, const upk_rotamer_t* R);
int upk_rotamer_nbr_slots(const upk_launch_t* L, const upk_rotamer_t* R);
/* 1-body energies -> node probabilities (rotamer.cpp:811-826, 239-256) */
int upk_rotamer_node_prob(const upk_launch_t* L, const upk_rotamer_t* R);
/* bead-pair energies accumulated into the residue-pair matrices (rotamer.cpp:829-846, interaction_graph.h:470-503) */
int upk_rotamer_pair_energy(const upk_launch_t* L, const upk_rotamer_t* R);
/* damped belief propagation + marginals (+ Bethe free energy): rotamer.cpp:1005-1061, 453-522, 283-302, 405-451, 854-866 */
int upk_rotamer_bp(const upk_launch_t* L, const upk_rotamer_t* R, int want_energy);
/* floats of exp(-E) pair matrices one workgroup of the cluster solve can keep in LDS (to choose bp_C) */
int upk_rotamer_bp_cluster_capacity(const upk_rotamer_t* R);
int upk_rotamer_bp_cluster_threads(void);
int upk_device_cu_count(void);   /* and the number of slots of one class it can own (one per lane) */
/* derivative push: pair marginal x pair gradient gathered per bead, node marginals to the 1-body parents
 * (rotamer.cpp:956-985, interaction_graph.h:525-555) */
int upk_rotamer_grad(const upk_launch_t* L, const upk_rotamer_t* R);

/* ---- parameter derivatives (the reference's get_param_deriv under -DPARAM_DERIV, deriv_engine.h:71-74) ------------
 * Each call handles ONE system of the batch and ADDS into `table`, which the caller has zeroed; layout = the node's
 * get_param().  They read the outputs, sensitivities and cached lists of the last evaluate_deriv.  Off the MD path. */
/* interaction_graph.h:404-416: hbond_coverage (quadspline coefficients), environment_coverage (zeros,
 * environment.cpp:62-65); sens_mode as in upk_igraph_grad */
int upk_igraph_param_deriv(const upk_launch_t* L, const upk_igraph_t* G, int system, int sens_mode, const float* sens1,
                           const float* sens2, long sens_sys_stride, int sens_stride, float* table);
/* rotamer.cpp:1064-1066: bead pairs weighted by the beliefs / pair marginals of the last solve */
int upk_rotamer_param_deriv(const upk_launch_t* L, const upk_rotamer_t* R, int system, float* table);
/* placement.cpp:144-160 (fixed placements: [n_layer][n_pos_dim]) */
int upk_placement_param_deriv(const upk_launch_t* L, const upk_placement_t* P, upk_coord_t aff, upk_coord_t out, int system, float* table);
/* environment.cpp:375-389 ([n_restype][n_coeff]) */
int upk_nonlinear_coupling_param_deriv(const upk_launch_t* L, upk_coord_t input, const int* types, int n_coeff, float offset,
                                       float inv_dx, int system, float* table);
/* sum over the elements of output component `comp` (hbond.cpp:436-448: n_hbond) */
int upk_column_sum(const upk_launch_t* L, upk_coord_t c, int comp, int system, float* out);

/* measured VALU issue ceilings of the device for the pair kernels' launch shape (one 1024-lane workgroup per CU): wave-level
 * fp32 FMA instructions per second, rates[0] for a dependent scalar chain, rates[1] with four independent chains per lane */
int upk_calibrate_valu(double* rates);

/* ---- optional restraint / external-field nodes (not emitted by the README configuration) ------------------------- */
/* single-atom potentials on pos; par is [n][8]:
 *   kind 0 atom_pos_spring (bonds.cpp:9-50)    x0[3], k
 *   kind 1 tension         (bonds.cpp:53-90)   tension_coeff[3]
 *   kind 2 AFM             (bonds.cpp:93-168)  k, starting_tip_pos[3], pulling_vel[3]; `time` = time_estimate
 *   kind 3 z_flat_bottom   (bonds.cpp:377-427) z0, radius, k
 * contrib: (n,3) contribution rows of pos' scatter plan */
int upk_point_potential(const upk_launch_t* L, int kind, upk_coord_t pos, const int* id, const float* par, int n, float time,
                        float* contrib, long contrib_stride, float* pot_terms);
/* contact (sidechain_radial.cpp:139-205): id (n,2); par [n][4] = energy, dist, 1/width, cutoff; contrib (n,2,3) */
int upk_contact(const upk_launch_t* L, upk_coord_t bead, const int* id, const float* par, int n, float* contrib, long contrib_stride,
                float* pot_terms);
/* constant (bonds.cpp:550-587), slice (bonds.cpp:589-621) */
int upk_broadcast_rows(const upk_launch_t* L, const float* value, upk_coord_t out);
int upk_slice_fwd(const upk_launch_t* L, upk_coord_t in, const int* id, upk_coord_t out);
int upk_slice_bwd(const upk_launch_t* L, upk_coord_t self, float* contrib, long contrib_stride);
/* uniform_transform (environment.cpp:158-235); jac [S][n_elem] */
int upk_uniform_transform_fwd(const upk_launch_t* L, upk_coord_t in, const float* coeff, int n_coeff, float offset, float inv_dx,
                              upk_coord_t out, float* jac);
int upk_uniform_transform_bwd(const upk_launch_t* L, upk_coord_t in, upk_coord_t self, const float* jac);
int upk_uniform_transform_param_deriv(const upk_launch_t* L, upk_coord_t in, const float* coeff, int n_coeff, float offset, float inv_dx,
                                      int system, float* table);
/* linear_coupling_uniform / linear_coupling_with_inactivation (environment.cpp:237-321) */
int upk_linear_coupling(const upk_launch_t* L, upk_coord_t in, const int* types, const float* couplings, upk_coord_t inact, int has_inact,
                        int inact_dim, float* pot_terms);
int upk_linear_coupling_param_deriv(const upk_launch_t* L, upk_coord_t in, const int* types, upk_coord_t inact, int has_inact, int inact_dim,
                                    int system, float* table);
/* membrane_potential (membrane_potential.cpp:13-155) */
typedef struct {
    int n_res, n_donor;
    const int *cb_index, *env_index, *restype;         /* [n_res] */
    const float *cov_midpoint, *cov_sharpness;          /* [n_restype] */
    const float *cb_coeff, *cb_table; int cb_nx;        /* monomial pieces [n_restype][cb_nx-1][4], raw table [n_restype][cb_nx] */
    const float *uhb_coeff, *uhb_table; int uhb_nx;     /* 2 layers: unpaired donor, unpaired acceptor */
    float cb_z_shift, cb_z_scale, uhb_z_shift, uhb_z_scale;
} upk_membrane_t;
int upk_membrane(const upk_launch_t* L, const upk_membrane_t* M, upk_coord_t cb, upk_coord_t env, upk_coord_t hb, float* pot_terms);

/* protein_hbond finish (src/hbond.cpp:320-335): copy the 6 infer components, out[6] = 1 - exp(-sum) */
int upk_protein_hbond_finish(const upk_launch_t* L, upk_coord_t infer, upk_coord_t out);
/* protein_hbond backward prologue/epilogue (src/hbond.cpp:343-365): sens_scaled and pass-through */
int upk_protein_hbond_bwd_pre(const upk_launch_t* L, upk_coord_t self, float* sens_scaled);
int upk_protein_hbond_passthrough(const upk_launch_t* L, upk_coord_t self, upk_coord_t infer, const int* loc1, int n1,
                                  const int* loc2, int n2);

/* device-side Metropolis for one swap set of replica exchange (src/main.cpp:251-273).  draw0 = number of
 * uniform 4-vectors already consumed from this round's generator; accepted has n_pair+1 entries, the last
 * one receives the generator position after this set. */
int upk_replica_swap(const upk_launch_t* L, upk_coord_t pos, const float* energy, const float* beta, int n_pair,
                     const int* pairs, uint32_t seed, uint64_t round, int draw0, int* accepted);
/* replica exchange across GPUs (src/main.cpp:227-275 over a ladder spread across ranks; host side: csrc/comm_rccl.cpp).
 * Everything is device-resident and stream-ordered between the RCCL calls: total potentials (node order, fp32),
 * Metropolis verdicts over the all-gathered energies (identical on every rank; accepted pairs trade their gathered
 * energies; draw_io = generator position within the attempt), and the coordinate moves (plan[p] = {kind, a, b}: 1 = both
 * local, swap; 2 = local system a takes staging row b, the partner's coordinates received from its rank). */
int upk_sum_potentials(const upk_launch_t* L, const float* const* node_pot, int n_node, float* out);
int upk_replica_decide(const upk_launch_t* L, float* energy_all, const float* beta_all, int n_pair, const int* pairs,
                       uint32_t seed, uint64_t round, int* draw_io, int* accepted);
int upk_replica_apply(const upk_launch_t* L, upk_coord_t pos, int n_pair, const int* plan, const int* accepted, const float* staging);
/* swap the coordinates of n_pair disjoint (s1, s2) pairs of systems; pairs is a device array */
int upk_swap_system_pairs(const upk_launch_t* L, upk_coord_t pos, int n_pair, const int* pairs);

#ifdef __cplusplus
}
#endif
#endif
