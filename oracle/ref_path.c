/*
 * ORACLE — TEST INFRASTRUCTURE ONLY (see ref_core.h header: PARITY UNPINNED).
 *
 * Restatement of the reference's hot-path DRIVERS, op for op in the reference's order:
 *   PowersDag::configure            common/apsu/powers.cpp:22-107
 *   create_powers_set               common/apsu/util/utils.cpp:146-177
 *   Receiver::ComputePowers         receiver/apsu/receiver_osn.cpp:395-488
 *   BatchedPlaintextPolyn::eval     receiver/apsu/bin_bundle.cpp:106-174
 *   ...::eval_patstock              receiver/apsu/bin_bundle.cpp:192-360
 * plus the DB-side layout rule of the BatchedPlaintextPolyn ctor (bin_bundle.cpp:366-430).
 */
#include "ref_path.h"
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ powers */

/* utils.cpp:146-177 ; returns count, fills out[] ascending */
int ref_create_powers_set(uint32_t ps_low_degree, uint32_t target_degree, uint32_t *out, int cap)
{
    if (ps_low_degree > target_degree || !target_degree) return -1;
    int k = 0;
    if (ps_low_degree) {
        for (uint32_t p = 1; p <= ps_low_degree; p++) { if (k < cap) out[k] = p; k++; }
        uint32_t first = ps_low_degree + 1, last = (target_degree / first) * first;
        for (uint32_t p = first; p <= last; p += first) { if (k < cap) out[k] = p; k++; }
    } else {
        for (uint32_t p = 1; p <= target_degree; p++) { if (k < cap) out[k] = p; k++; }
    }
    return k;
}

static int in_set(const uint32_t *s, int n, uint32_t v)
{
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (s[mid] == v) return mid;
        if (s[mid] < v) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

/* powers.cpp:22-107.  sources/targets ascending.  nodes[] gets one entry per target in
   ascending power order.  Returns depth or -1. */
int ref_powers_dag_configure(const uint32_t *sources, int ns, const uint32_t *targets, int nt,
                             ref_dag_node *nodes)
{
    if (in_set(sources, ns, 0) >= 0 || in_set(sources, ns, 1) < 0) return -1;
    if (in_set(targets, nt, 0) >= 0 || in_set(targets, nt, 1) < 0) return -1;
    for (int i = 0; i < ns; i++) if (in_set(targets, nt, sources[i]) < 0) return -1;
    uint32_t max_t = targets[nt - 1];
    uint32_t *depth = (uint32_t *)calloc((size_t)max_t + 1, sizeof(uint32_t));
    int curr_depth = 0;
    for (int ti = 0; ti < nt; ti++) {
        uint32_t cp = targets[ti];
        nodes[ti].power = cp;
        if (in_set(sources, ns, cp) >= 0) {
            nodes[ti].depth = 0; nodes[ti].p1 = nodes[ti].p2 = 0;
            depth[cp] = 0;
            continue;
        }
        uint32_t od = cp - 1, os1 = cp - 1, os2 = 1;
        for (int si = 0; si < nt; si++) {
            uint32_t s1 = targets[si];
            if (s1 >= cp) break;
            uint32_t s2 = cp - s1;
            if (in_set(targets, nt, s2) < 0) continue;
            uint32_t d = (depth[s1] > depth[s2] ? depth[s1] : depth[s2]) + 1;
            if (d < od) { od = d; os1 = s1; os2 = s2; }
        }
        nodes[ti].depth = od; nodes[ti].p1 = os1; nodes[ti].p2 = os2;
        depth[cp] = od;
        if ((int)od > curr_depth) curr_depth = (int)od;
    }
    free(depth);
    return curr_depth;
}

static size_t ct_words(const ref_ctx *c, int polys, int chain_idx)
{
    return (size_t)polys * (chain_idx + 1) * c->n;
}

/* Size of every target power: 2 with key switching (each product is relinearised, receiver_osn.cpp:430-432); without it
   (one coefficient prime) a product has size(parent1) + size(parent2) - 1.  sizes[] is indexed by power (entries of powers
   that are no targets stay untouched).  Returns the largest size, or -1 where SEAL's multiply throws (above 16). */
int ref_power_sizes(const ref_ctx *c, const ref_dag_node *nodes, int n_nodes, uint32_t *sizes)
{
    int max_depth = 0, max_size = 2;
    for (int i = 0; i < n_nodes; i++) if ((int)nodes[i].depth > max_depth) max_depth = (int)nodes[i].depth;
    for (int i = 0; i < n_nodes; i++) if (!nodes[i].depth) sizes[nodes[i].power] = 2;
    for (int d = 1; d <= max_depth; d++)
        for (int i = 0; i < n_nodes; i++) {
            if ((int)nodes[i].depth != d) continue;
            uint32_t s = c->using_keyswitching ? 2 : sizes[nodes[i].p1] + sizes[nodes[i].p2] - 1;
            if (s > REF_CT_SIZE_MAX) return -1;
            sizes[nodes[i].power] = s;
            if ((int)s > max_size) max_size = (int)s;
        }
    return max_size;
}

/* receiver_osn.cpp:395-488.
   powers[p] (p = power index, 0 unused): caller-allocated buffers for every target power, 3 * first_L * n words each with key
   switching, REF_CT_SIZE_MAX * first_L * n without; source powers hold size-2 coefficient-form cts at the first data level.
   On return each target power is (per :459-487):
     ps_low_degree == 0     : level high, NTT form
     power <= ps_low_degree : level low,  NTT form
     power >  ps_low_degree : level high, coefficient form
   of size 2 with key switching, of size sizes[p] (ref_power_sizes; required then) without.  -3: SEAL throws (size > 16). */
int ref_compute_powers(const ref_ctx *c, uint64_t **powers, const ref_dag_node *nodes, int n_nodes,
                       const uint64_t *rk, uint32_t ps_low_degree, uint32_t *sizes)
{
    int first = c->first_chain_idx;
    int max_depth = 0;
    for (int i = 0; i < n_nodes; i++) if ((int)nodes[i].depth > max_depth) max_depth = (int)nodes[i].depth;
    const int relinearize = c->using_keyswitching;                                        /* :416 */
    if (!relinearize) {
        if (sizes) { if (ref_power_sizes(c, nodes, n_nodes, sizes) < 0) return -3; }
        else if (max_depth > 0) return -1;
    }
    /* parallel_apply visits a node once both parents are done; any topological order gives
       the same values.  Visit by depth. */
    /* the reference runs one task per DAG node on its thread pool (powers.h:158-278); nodes of equal depth
       are independent, so they are the OpenMP work items here (OMP_NUM_THREADS = the reference's -t) */
    for (int d = 1; d <= max_depth; d++) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int i = 0; i < n_nodes; i++) {
            if ((int)nodes[i].depth != d) continue;
            const ref_dag_node *nd = &nodes[i];
            uint64_t *prod = powers[nd->power];
            if (relinearize) {
                if (nd->p1 == nd->p2) ref_square(c, powers[nd->p1], prod, first);        /* :422 */
                else ref_multiply(c, powers[nd->p1], powers[nd->p2], prod, first);       /* :424 */
                ref_relinearize(c, prod, rk, first);                                     /* :431 */
            } else {
                ref_multiply_sized(c, powers[nd->p1], (int)sizes[nd->p1], powers[nd->p2], (int)sizes[nd->p2], prod, first);
            }
        }
    }
    int high = ref_clamp_chain_idx(c, 1), low = ref_clamp_chain_idx(c, 2);                /* :451-454 */
#pragma omp parallel for schedule(dynamic, 1)                                             /* :458-487 one task per power */
    for (int i = 0; i < n_nodes; i++) {
        uint32_t power = nodes[i].power;
        uint64_t *ct = powers[power];
        int polys = (!relinearize && sizes) ? (int)sizes[power] : 2;
        int lvl = first;
        int target = (!ps_low_degree || power > ps_low_degree) ? high : low;
        while (lvl > target) { ref_mod_switch_to_next(c, ct, polys, lvl); lvl--; }        /* :463,471,478 */
        if (!ps_low_degree || power <= ps_low_degree) ref_transform_to_ntt(c, ct, polys, lvl); /* :467,475 */
    }
    return 0;
}

/* bin_bundle.cpp:385-389,418-420: which degree indices are stored in NTT form, and at what level */
int ref_plain_chain_idx(const ref_ctx *c, uint32_t ps_low_degree)
{
    int v = ps_low_degree ? 2 : 1;
    return c->first_chain_idx < v ? c->first_chain_idx : v;
}
int ref_coeff_is_ntt(uint32_t ps_low_degree, uint32_t i)
{
    return (!ps_low_degree && i != 0) || (ps_low_degree && (i % (ps_low_degree + 1)) != 0);
}

/* Plaintext::unsafe_load(...) at bin_bundle.cpp:143,252,281,315,329,340 deserialises (copies)
   the stored plaintext into a scratch Plaintext on every use; restated as a memcpy so the CPU
   baseline pays the same memory traffic. */
static const uint64_t *load_coeff(uint64_t *scratch, const uint64_t *stored, size_t words)
{
    memcpy(scratch, stored, words * sizeof(uint64_t));
    return scratch;
}

/* bin_bundle.cpp:106-174.  powers[d] for d=1..degree at level `lvl` in NTT form.
   coeffs[0]: n words mod t (coefficient form); coeffs[d>0]: (lvl+1)*n words NTT form.
   sizes: polynomials of every power (NULL: 2 each, the case with key switching).  out: *out_size * n words at the last level
   (capacity: 2 * n words without sizes, REF_CT_SIZE_MAX * n with).  Returns 0 / -1 (not enough powers). */
int ref_eval(const ref_ctx *c, uint64_t *const *powers, int n_powers, const uint64_t *const *coeffs,
             int n_coeffs, int lvl, const uint64_t *mask, uint64_t *out, const uint32_t *sizes, uint32_t *out_size)
{
    if (n_powers < (n_coeffs > 2 ? n_coeffs : 2)) return -1;                              /* :116-118 */
    const int cap = sizes ? REF_CT_SIZE_MAX : 2;
    size_t n = c->n, w = ct_words(c, cap, lvl);
    uint64_t *result = (uint64_t *)calloc(w, sizeof(uint64_t));                           /* :132-134 size 2, zero */
    int rs = 2;
    uint64_t *temp = (uint64_t *)malloc(w * sizeof(uint64_t));
    uint64_t *scratch = (uint64_t *)malloc((size_t)(lvl + 1) * n * sizeof(uint64_t));
    for (int deg = 1; deg < n_coeffs; deg++) {
        const int ps = sizes ? (int)sizes[deg] : 2;
        const uint64_t *co = load_coeff(scratch, coeffs[deg], (size_t)(lvl + 1) * n);     /* :143 */
        ref_multiply_plain_ntt(c, powers[deg], co, temp, ps, lvl);                        /* :147 */
        ref_add(c, result, temp, ps, lvl);       /* :148 add_inplace: the longer operand's extra polynomials are copied = added to zero */
        if (ps > rs) rs = ps;
    }
    ref_transform_from_ntt(c, result, rs, lvl);                                           /* :154 */
    ref_add_plain(c, result, load_coeff(scratch, coeffs[0], n), n, lvl);                  /* :159 */
    ref_add_plain(c, result, mask, n, lvl);                                               /* :162 */
    while (lvl > 0) { ref_mod_switch_to_next(c, result, rs, lvl); lvl--; }                /* :168-170 */
    ref_clear_irrelevant_bits(c, result, rs);                                             /* :171 */
    memcpy(out, result, (size_t)rs * n * sizeof(uint64_t));
    if (out_size) *out_size = (uint32_t)rs;
    free(result); free(temp); free(scratch);
    return 0;
}

/* bin_bundle.cpp:192-360; sizes / out / out_size as in ref_eval.  -3: a product would exceed SEAL's largest ciphertext. */
int ref_eval_patstock(const ref_ctx *c, uint64_t *const *powers, int n_powers,
                      const uint64_t *const *coeffs, int n_coeffs, uint32_t ps_low_degree,
                      const uint64_t *rk, const uint64_t *mask, uint64_t *out, const uint32_t *sizes, uint32_t *out_size)
{
    if (n_powers < (n_coeffs > 2 ? n_coeffs : 2)) return -1;                              /* :204-206 */
    size_t degree = (size_t)n_coeffs - 1;
    if (ps_low_degree <= 1 || ps_low_degree >= degree) return -2;                         /* :209-213 */
    /* Without key switching (one coefficient prime) the reference skips relinearize_inplace (:308-310): the result keeps
       the size of the largest product (at least 3, :238-240), and the powers themselves are longer than 2 (`sizes`). */
    if (!c->using_keyswitching && !sizes) return -1;
    int high = ref_clamp_chain_idx(c, 1);                                                 /* :220 */
    int low = ref_plain_chain_idx(c, ps_low_degree);      /* level of low powers & NTT plaintexts */
    size_t n = c->n;
    size_t h = (size_t)ps_low_degree + 1, H = degree / h;                                 /* :225-227 */
    const int cap = sizes ? REF_CT_SIZE_MAX : 3;
#define SZ(p) (sizes ? (int)sizes[p] : 2)
    size_t wl = ct_words(c, cap, low), wh = ct_words(c, cap, high);
    size_t ptw = (size_t)(low + 1) * n;

    uint64_t *result = (uint64_t *)calloc(wh, sizeof(uint64_t));                          /* :238-240 size 3, zero */
    int rs = 3;
    uint64_t *temp = (uint64_t *)malloc((wl > wh ? wl : wh) * sizeof(uint64_t));
    uint64_t *temp_in = (uint64_t *)malloc((wl > wh ? wl : wh) * sizeof(uint64_t));
    uint64_t *prod = (uint64_t *)malloc(wh * sizeof(uint64_t));
    uint64_t *scratch = (uint64_t *)malloc(ptw * sizeof(uint64_t));
    int rc = 0;

    for (size_t i = 1; i <= H && !rc; i++) {                                              /* :248-304 */
        size_t jmax = (i < H) ? h - 1 : degree % h;
        if (i == H && jmax == 0) break;                                                   /* :279 */
        int s_in = 2;
        if (sizes) memset(temp_in, 0, wl * sizeof(uint64_t));      /* later terms may be longer than the first (add_inplace copies) */
        for (size_t j = 1; j <= jmax; j++) {
            const uint64_t *co = load_coeff(scratch, coeffs[i * h + j], ptw);             /* :252,281 */
            ref_multiply_plain_ntt(c, powers[j], co, temp, SZ(j), low);                   /* :258,287 */
            if (j == 1) memcpy(temp_in, temp, ct_words(c, SZ(1), low) * sizeof(uint64_t));
            else ref_add(c, temp_in, temp, SZ(j), low);                                   /* :264,293 */
            if (SZ(j) > s_in) s_in = SZ(j);
        }
        ref_transform_from_ntt(c, temp_in, s_in, low);                                    /* :268,297 */
        for (int l = low; l > high; l--) ref_mod_switch_to_next(c, temp_in, s_in, l);     /* :269,298 */
        const int sp = s_in + SZ(i * h) - 1;
        if (ref_multiply_sized(c, temp_in, s_in, powers[i * h], SZ(i * h), prod, high)) { rc = -3; break; }   /* :272,301 */
        ref_add(c, result, prod, sp, high);                                               /* :273,303 */
        if (sp > rs) rs = sp;
    }
    if (rc) { free(result); free(temp); free(temp_in); free(prod); free(scratch); return rc; }
    if (c->using_keyswitching) { ref_relinearize(c, result, rk, high); rs = 2; }          /* :308-310 */

    for (size_t j = 1; j < h; j++) {                                                      /* :314-324 */
        const uint64_t *co = load_coeff(scratch, coeffs[j], ptw);
        ref_multiply_plain_ntt(c, powers[j], co, temp, SZ(j), low);
        ref_transform_from_ntt(c, temp, SZ(j), low);
        for (int l = low; l > high; l--) ref_mod_switch_to_next(c, temp, SZ(j), l);
        ref_add(c, result, temp, SZ(j), high);
        if (SZ(j) > rs) rs = SZ(j);
    }
    for (size_t i = 1; i <= H; i++) {                                                     /* :328-337 */
        const uint64_t *co = load_coeff(scratch, coeffs[i * h], n);
        ref_multiply_plain_coeff(c, powers[i * h], co, n, temp, SZ(i * h), high);
        ref_add(c, result, temp, SZ(i * h), high);
        if (SZ(i * h) > rs) rs = SZ(i * h);
    }
    ref_add_plain(c, result, load_coeff(scratch, coeffs[0], n), n, high);                 /* :345 */
    ref_add_plain(c, result, mask, n, high);                                              /* :346 */
    for (int l = high; l > 0; l--) ref_mod_switch_to_next(c, result, rs, l);              /* :354-356 */
    ref_clear_irrelevant_bits(c, result, rs);                                             /* :357 */
    memcpy(out, result, (size_t)rs * n * sizeof(uint64_t));
    if (out_size) *out_size = (uint32_t)rs;
#undef SZ
    free(result); free(temp); free(temp_in); free(prod); free(scratch);
    return 0;
}
