/*
 * TEST INFRASTRUCTURE ONLY - CPU restatement of the integer half of the AIG-GAN hot path.
 *
 * Nothing under gan_amd/ may call into this file: it exists so that tests, the
 * smoke check and the cpu_baseline leg of bench.py have an independent checker
 * for the HIP kernels.  Parity status: PINNED - every function here is checked
 * against golden vectors captured from the reference itself (tests/golden/,
 * tests/test_oracle_golden.py).
 *
 * Each function follows the reference lines cited above it; sequential scalar
 * code, one thread.  Build: `make -C oracle` (gcc only).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define PI_T 0
#define PO_T 1
#define AND_T 2

/* ------------------------------------------------------------------------------------------
 * Processing order: models/generator.py:70-82 (_build_order).
 * order = indices sorted by (depth[i], i); without a usable depth vector the key is
 * 0 for a PI and i otherwise.  `depth` may be NULL.  Python compares the keys as numbers,
 * so -0.0 == 0.0 and int/float depths order alike; doubles carry both.
 * ------------------------------------------------------------------------------------------ */
typedef struct { double key; int64_t idx; } okey_t;

static int okey_cmp(const void *a, const void *b) {
    const okey_t *x = (const okey_t *)a, *y = (const okey_t *)b;
    if (x->key < y->key) return -1;
    if (x->key > y->key) return 1;
    return (x->idx > y->idx) - (x->idx < y->idx);
}

int orc_build_order(int64_t n, const int64_t *types, const double *depth, int64_t *order, int64_t *rank) {
    okey_t *k = (okey_t *)malloc(sizeof(okey_t) * (size_t)(n > 0 ? n : 1));
    if (!k) return -1;
    for (int64_t i = 0; i < n; ++i) {
        k[i].idx = i;
        k[i].key = depth ? depth[i] : (types[i] == PI_T ? 0.0 : (double)i);
    }
    qsort(k, (size_t)n, sizeof(okey_t), okey_cmp);
    for (int64_t r = 0; r < n; ++r) { order[r] = k[r].idx; if (rank) rank[k[r].idx] = r; }
    free(k);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Candidate windows: models/generator.py:131-158.
 * Walk `order`; a PI only becomes seen.  For any other node v (a "target"):
 *   cand = ascending ids u with seen[u] && type[u] != PO && u != v, last K of them kept;
 *   no candidate at all -> fallback source = largest seen id (any type), if something was seen.
 * Outputs (targets numbered in processing order):
 *   tgt[t]            node id of target t
 *   cand[t*K .. +cnt) ascending candidate ids, cnt[t] of them (0 when the fallback applies)
 *   fb[t]             fallback source id, or -1
 * Returns the number of targets.
 * ------------------------------------------------------------------------------------------ */
int64_t orc_candidates(int64_t n, const int64_t *types, const int64_t *order, int64_t K,
                       int64_t *tgt, int64_t *cand, int64_t *cnt, int64_t *fb) {
    /* Only the K largest seen non-PO ids can ever enter a window (generator.py:157-158 keeps cand_idx[-K:] of the
     * ascending id list), so that is all that is kept: `top` holds them ascending, `ns` counts every seen one. */
    int64_t cap = K > 0 ? K : 1;
    int64_t *top = (int64_t *)malloc(sizeof(int64_t) * (size_t)cap);
    int64_t ntop = 0, ns = 0, max_seen = -1, T = 0;
    if (!top) return -1;
    for (int64_t r = 0; r < n; ++r) {
        int64_t v = order[r];
        if (types[v] != PI_T) {
            int64_t c = ns < K ? ns : K;
            tgt[T] = v;
            cnt[T] = c;
            fb[T] = (c == 0) ? max_seen : -1;
            for (int64_t j = 0; j < c; ++j) cand[T * K + j] = top[ntop - c + j];
            ++T;
        }
        if (v > max_seen) max_seen = v;
        if (types[v] != PO_T) { /* becomes a possible source */
            ++ns;
            if (K > 0 && (ntop < K || v > top[0])) {
                int64_t lo = 0, hi = ntop;
                while (lo < hi) { int64_t m = (lo + hi) / 2; if (top[m] < v) lo = m + 1; else hi = m; }
                if (ntop < K) {          /* room left: shift the tail up */
                    memmove(top + lo + 1, top + lo, sizeof(int64_t) * (size_t)(ntop - lo));
                    top[lo] = v;
                    ++ntop;
                } else {                 /* full: the smallest drops out, everything below the slot shifts down */
                    memmove(top, top + 1, sizeof(int64_t) * (size_t)(lo - 1));
                    top[lo - 1] = v;
                }
            }
        }
    }
    free(top);
    return T;
}

/* ------------------------------------------------------------------------------------------
 * pos[i]: models/constraints.py:118-132 (type_order_index) - rank in (all PI, all AND, all PO),
 * index order inside each group.  The reference leaves pos of any other type uninitialised
 * (torch.empty); here such nodes are placed after every PO so the result is defined.
 * ------------------------------------------------------------------------------------------ */
void orc_type_order_index(int64_t n, const int64_t *types, int64_t *pos) {
    int64_t r = 0;
    for (int64_t i = 0; i < n; ++i) if (types[i] == PI_T) pos[i] = r++;
    for (int64_t i = 0; i < n; ++i) if (types[i] == AND_T) pos[i] = r++;
    for (int64_t i = 0; i < n; ++i) if (types[i] == PO_T) pos[i] = r++;
    for (int64_t i = 0; i < n; ++i) if (types[i] != PI_T && types[i] != AND_T && types[i] != PO_T) pos[i] = r++;
}

/* open-addressing set of (u,v) pairs -> slot payload */
typedef struct { int64_t u, v, val; } pslot_t;
typedef struct { pslot_t *s; uint64_t mask; } pset_t;

static int pset_init(pset_t *h, int64_t cap_hint) {
    uint64_t c = 16;
    while (c < (uint64_t)(cap_hint * 2 + 1)) c <<= 1;
    h->s = (pslot_t *)malloc(sizeof(pslot_t) * c);
    if (!h->s) return -1;
    for (uint64_t i = 0; i < c; ++i) h->s[i].val = -1;
    h->mask = c - 1;
    return 0;
}
static pslot_t *pset_find(pset_t *h, int64_t u, int64_t v) {
    uint64_t x = (uint64_t)u * 0x9E3779B97F4A7C15ull ^ ((uint64_t)v + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full;
    x ^= x >> 29;
    for (uint64_t i = x & h->mask;; i = (i + 1) & h->mask) {
        pslot_t *p = &h->s[i];
        if (p->val < 0 || (p->u == u && p->v == v)) return p;
    }
}

/* breadth-first levels from every node with start[i] != 0 over CSR (ptr, adj); -1 = not reached.
 * models/constraints.py:88-112 re-enqueues on improvement; the fixed point is the BFS level. */
static void bfs_levels(int64_t n, const int64_t *ptr, const int64_t *adj, const uint8_t *start, int64_t *lev) {
    int64_t *q = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    int64_t head = 0, tail = 0;
    for (int64_t i = 0; i < n; ++i) { lev[i] = -1; if (start[i]) { lev[i] = 0; q[tail++] = i; } }
    while (head < tail) {
        int64_t u = q[head++];
        for (int64_t e = ptr[u]; e < ptr[u + 1]; ++e) {
            int64_t v = adj[e];
            if (lev[v] < 0) { lev[v] = lev[u] + 1; q[tail++] = v; }
        }
    }
    free(q);
}

static void build_csr(int64_t n, int64_t e, const int64_t *from, const int64_t *to, int64_t *ptr, int64_t *adj) {
    memset(ptr, 0, sizeof(int64_t) * (size_t)(n + 1));
    for (int64_t i = 0; i < e; ++i) ptr[from[i] + 1]++;
    for (int64_t i = 0; i < n; ++i) ptr[i + 1] += ptr[i];
    int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    memcpy(cur, ptr, sizeof(int64_t) * (size_t)n);
    for (int64_t i = 0; i < e; ++i) adj[cur[from[i]]++] = to[i];
    free(cur);
}

/* ------------------------------------------------------------------------------------------
 * Repair: models/constraints.py:508-669 (enforce_aig_constraints), strict=False.
 *   n, types[n]         node types (x[:,0].long())
 *   e, src/dst/inv[e]   edges and int(edge_attr)
 *   depth               node_depth as doubles, or NULL when absent / of the wrong length
 * Outputs: out_src/out_dst/out_inv (capacity 2n), *out_e; node_depth[n] (BFS level from the PIs
 * over the cleaned, pre-repair edge list; -1 unreachable); inv_cnt[n] = min(2, sum of inversion
 * bits entering each node over that same list).
 * Cleaning (:525-546): drop self loops, merge duplicate (u,v) keeping the first with OR-ed
 * inversion (:15-50), drop out-of-range endpoints, drop edges with pos[u] >= pos[v].
 * Selection (:558-620): PO keeps the legal predecessor minimising (-is_and, -int(depth), u);
 * AND keeps the two smallest-pos legal predecessors, first widening the pool with every other
 * PI/AND node (inversion 0) when it has fewer than two; emitted grouped by target id.
 * ------------------------------------------------------------------------------------------ */
int orc_enforce(int64_t n, const int64_t *types, int64_t e, const int64_t *src, const int64_t *dst,
                const int64_t *inv, const double *depth,
                int64_t *out_src, int64_t *out_dst, int64_t *out_inv, int64_t *out_e,
                int64_t *node_depth, int64_t *inv_cnt) {
    *out_e = 0;
    if (e == 0) { /* :517-522 */
        for (int64_t i = 0; i < n; ++i) { node_depth[i] = -1; inv_cnt[i] = 0; }
        return 0;
    }
    int64_t *cs = (int64_t *)malloc(sizeof(int64_t) * (size_t)e * 3);
    int64_t *cd = cs + e, *ci = cd + e;
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    pset_t hs;
    if (!cs || !pos || pset_init(&hs, e)) return -1;
    int64_t m = 0;
    for (int64_t i = 0; i < e; ++i) { /* remove_self_and_dups */
        int64_t u = src[i], v = dst[i];
        if (u == v) continue;
        pslot_t *p = pset_find(&hs, u, v);
        if (p->val >= 0) { if (ci[p->val] == 0 && inv[i] == 1) ci[p->val] = 1; continue; }
        p->u = u; p->v = v; p->val = m;
        cs[m] = u; cd[m] = v; ci[m] = inv[i]; ++m;
    }
    free(hs.s);
    orc_type_order_index(n, types, pos);
    int64_t k = 0;
    for (int64_t i = 0; i < m; ++i) { /* range filter, then forward filter */
        int64_t u = cs[i], v = cd[i];
        if (u < 0 || u >= n || v < 0 || v >= n) continue;
        if (!(pos[u] < pos[v])) continue;
        cs[k] = u; cd[k] = v; ci[k] = ci[i]; ++k;
    }
    m = k;
    /* predecessors by target, edge order kept */
    int64_t *ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *eid = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    int64_t *ids = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    for (int64_t i = 0; i < m; ++i) ids[i] = i;
    build_csr(n, m, cd, ids, ptr, eid);
    /* the three smallest-pos PI/AND nodes (widening pool of :585-594 sorted by pos) */
    int64_t small[3] = {-1, -1, -1}, nsmall = 0, n_src_nodes = 0;
    {
        int64_t *bypos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
        for (int64_t i = 0; i < n; ++i) bypos[i] = -1;
        for (int64_t i = 0; i < n; ++i) if (types[i] == PI_T || types[i] == AND_T) { bypos[pos[i]] = i; ++n_src_nodes; }
        for (int64_t r = 0; r < n && nsmall < 3; ++r) if (bypos[r] >= 0) small[nsmall++] = bypos[r];
        free(bypos);
    }
    int64_t oe = 0;
    for (int64_t v = 0; v < n; ++v) {
        int64_t t = types[v];
        if (t == PO_T) {
            int64_t best = -1; int64_t b_and = 0, b_dep = 0;
            for (int64_t q = ptr[v]; q < ptr[v + 1]; ++q) {
                int64_t id = eid[q], u = cs[id];
                if (!(types[u] == PI_T || types[u] == AND_T) || u == v) continue;
                int64_t is_and = types[u] == AND_T;
                int64_t dep = depth ? (int64_t)depth[u] : 0; /* int(): truncation toward zero */
                int better = 0;
                if (best < 0) better = 1;
                else if (is_and != b_and) better = is_and > b_and;
                else if (dep != b_dep) better = dep > b_dep;
                else better = u < cs[best];
                if (better) { best = id; b_and = is_and; b_dep = dep; }
            }
            if (best >= 0) { out_src[oe] = cs[best]; out_dst[oe] = v; out_inv[oe] = ci[best]; ++oe; }
        } else if (t == AND_T) {
            int64_t b1 = -1, b2 = -1, nl = 0; /* legal predecessors with the two smallest pos */
            for (int64_t q = ptr[v]; q < ptr[v + 1]; ++q) {
                int64_t id = eid[q], u = cs[id];
                if (!(types[u] == PI_T || types[u] == AND_T) || u == v) continue;
                ++nl;
                if (b1 < 0 || pos[u] < pos[cs[b1]]) { b2 = b1; b1 = id; }
                else if (b2 < 0 || pos[u] < pos[cs[b2]]) b2 = id;
            }
            if (nl >= 2) {
                out_src[oe] = cs[b1]; out_dst[oe] = v; out_inv[oe] = ci[b1]; ++oe;
                out_src[oe] = cs[b2]; out_dst[oe] = v; out_inv[oe] = ci[b2]; ++oe;
            } else {
                /* pool = the legal one (if any) + every other PI/AND node except v; two smallest pos */
                int64_t pick[2], np = 0;
                for (int64_t j = 0; j < nsmall && np < 2; ++j) if (small[j] != v) pick[np++] = small[j];
                int64_t pool = n_src_nodes - 1; /* v itself is an AND, hence a source node */
                if (pool >= 2 && np == 2) {
                    for (int64_t j = 0; j < 2; ++j) {
                        int64_t u = pick[j];
                        out_src[oe] = u; out_dst[oe] = v;
                        out_inv[oe] = (nl == 1 && cs[b1] == u) ? ci[b1] : 0;
                        ++oe;
                    }
                }
            }
        }
    }
    *out_e = oe;
    /* depth + inversion counts from the cleaned pre-repair list (:641-661) */
    {
        int64_t *fptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
        int64_t *fadj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
        uint8_t *start = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
        build_csr(n, m, cs, cd, fptr, fadj);
        for (int64_t i = 0; i < n; ++i) start[i] = types[i] == PI_T;
        bfs_levels(n, fptr, fadj, start, node_depth);
        free(fptr); free(fadj); free(start);
    }
    for (int64_t i = 0; i < n; ++i) inv_cnt[i] = 0;
    for (int64_t i = 0; i < m; ++i) inv_cnt[cd[i]] += ci[i];
    for (int64_t i = 0; i < n; ++i) if (inv_cnt[i] > 2) inv_cnt[i] = 2;
    free(cs); free(pos); free(ptr); free(eid); free(ids);
    return 0;
}

static int i64_cmp(const void *a, const void *b) {
    int64_t x = *(const int64_t *)a, y = *(const int64_t *)b;
    return (x > y) - (x < y);
}

/* nodes on some PI -> PO walk = reached forward from the PIs and backward from the POs */
static int64_t on_path_count(int64_t n, const int64_t *types, int64_t e, const int64_t *src, const int64_t *dst,
                             uint8_t *on_path) {
    int64_t *ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *adj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(e > 0 ? e : 1));
    int64_t *lf = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * 2), *lr = lf + n;
    uint8_t *st = (uint8_t *)calloc((size_t)(n > 0 ? n : 1), 1);
    build_csr(n, e, src, dst, ptr, adj);
    for (int64_t i = 0; i < n; ++i) st[i] = types[i] == PI_T;
    bfs_levels(n, ptr, adj, st, lf);
    build_csr(n, e, dst, src, ptr, adj);
    for (int64_t i = 0; i < n; ++i) st[i] = types[i] == PO_T;
    bfs_levels(n, ptr, adj, st, lr);
    int64_t c = 0;
    for (int64_t i = 0; i < n; ++i) { int ok = lf[i] >= 0 && lr[i] >= 0; if (on_path) on_path[i] = (uint8_t)ok; c += ok; }
    free(ptr); free(adj); free(lf); free(st);
    return c;
}

/* ------------------------------------------------------------------------------------------
 * Violation counts: models/constraints.py:253-366 (aig_constraint_penalties).
 * out[0..8] = pi_in, po_out, backedge, selfloop, dup_fanin, indeg_mse, fanin_type,
 *             reach_violation, total (weights 1,1,2,3,1,1,1,0.5).  All endpoints in range.
 * ------------------------------------------------------------------------------------------ */
int orc_penalties(int64_t n, const int64_t *types, int64_t e, const int64_t *src, const int64_t *dst, double *out) {
    for (int i = 0; i < 9; ++i) out[i] = 0.0;
    if (e == 0) return 0;
    int64_t *pos = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *indeg = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    int64_t *pairs = (int64_t *)malloc(sizeof(int64_t) * (size_t)e);
    orc_type_order_index(n, types, pos);
    int64_t pi_in = 0, po_out = 0, back = 0, self = 0, ftype = 0;
    for (int64_t i = 0; i < e; ++i) {
        int64_t u = src[i], v = dst[i];
        self += u == v;
        pi_in += types[v] == PI_T;
        po_out += types[u] == PO_T;
        back += !(pos[u] < pos[v]);
        ftype += (types[v] == AND_T || types[v] == PO_T) && !(types[u] == PI_T || types[u] == AND_T);
        indeg[v]++;
        pairs[i] = u * n + v;
    }
    qsort(pairs, (size_t)e, sizeof(int64_t), i64_cmp);
    int64_t uniq = 1;
    for (int64_t i = 1; i < e; ++i) uniq += pairs[i] != pairs[i - 1];
    int64_t mse = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t target = types[i] == PO_T ? 1 : (types[i] == AND_T ? 2 : 0);
        int64_t d = indeg[i] - target;
        mse += d < 0 ? -d : d;
    }
    int64_t onp = on_path_count(n, types, e, src, dst, NULL);
    out[0] = (double)pi_in; out[1] = (double)po_out; out[2] = (double)back; out[3] = (double)self;
    out[4] = (double)(e - uniq); out[5] = (double)mse; out[6] = (double)ftype; out[7] = (double)(n - onp);
    out[8] = 1.0 * out[0] + 1.0 * out[1] + 2.0 * out[2] + 3.0 * out[3] + 1.0 * out[4] + 1.0 * out[5]
           + 1.0 * out[6] + 0.5 * out[7];
    free(pos); free(indeg); free(pairs);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Strict validation: models/constraints.py:375-505 (validate_strict_aig).
 * Per-edge flag bits (eflag[e]):  1 self loop, 2 repeated (u,v) (every occurrence after the first),
 *   4 into a PI, 8 out of a PO, 16 source type not PI/AND for an AND/PO target.
 * Per-node code (ncode[n]): 0 fine, 1 unknown type, 2 PO with indegree != 1, 3 AND with
 *   indegree != 2, 4 AND with the same fan-in twice; indeg[n] as counted by the reference.
 * summary[0] = 1 if the graph has a cycle (Kahn, :64-85), summary[1] = nodes not on a PI->PO path.
 * on_path[n] marks covered nodes.
 * ------------------------------------------------------------------------------------------ */
int orc_validate(int64_t n, const int64_t *types, int64_t e, const int64_t *src, const int64_t *dst,
                 uint8_t *eflag, uint8_t *ncode, int64_t *indeg, uint8_t *on_path, int64_t *summary) {
    pset_t hs;
    if (pset_init(&hs, e)) return -1;
    for (int64_t i = 0; i < n; ++i) { indeg[i] = 0; ncode[i] = 0; }
    int64_t *first = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1) * 2), *second = first + n;
    for (int64_t i = 0; i < n; ++i) first[i] = second[i] = -1;
    for (int64_t i = 0; i < e; ++i) {
        int64_t u = src[i], v = dst[i];
        uint8_t f = 0;
        if (u == v) f |= 1;
        pslot_t *p = pset_find(&hs, u, v);
        if (p->val >= 0) f |= 2; else { p->u = u; p->v = v; p->val = 1; }
        if (types[v] == PI_T) f |= 4;
        if (types[u] == PO_T) f |= 8;
        if ((types[v] == AND_T || types[v] == PO_T) && !(types[u] == PI_T || types[u] == AND_T)) f |= 16;
        eflag[i] = f;
        if (indeg[v] == 0) first[v] = u; else if (indeg[v] == 1) second[v] = u;
        indeg[v]++;
    }
    free(hs.s);
    for (int64_t i = 0; i < n; ++i) {
        int64_t t = types[i];
        if (t != PI_T && t != PO_T && t != AND_T) ncode[i] = 1;
        else if (t == PO_T && indeg[i] != 1) ncode[i] = 2;
        else if (t == AND_T && indeg[i] != 2) ncode[i] = 3;
        else if (t == AND_T && first[i] == second[i]) ncode[i] = 4;
    }
    free(first);
    /* Kahn */
    {
        int64_t *ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
        int64_t *adj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(e > 0 ? e : 1));
        int64_t *deg = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
        int64_t *q = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
        build_csr(n, e, src, dst, ptr, adj);
        int64_t head = 0, tail = 0;
        for (int64_t i = 0; i < n; ++i) { deg[i] = indeg[i]; if (!deg[i]) q[tail++] = i; }
        while (head < tail) {
            int64_t u = q[head++];
            for (int64_t k = ptr[u]; k < ptr[u + 1]; ++k) if (--deg[adj[k]] == 0) q[tail++] = adj[k];
        }
        summary[0] = tail != n;
        free(ptr); free(adj); free(deg); free(q);
    }
    summary[1] = n - on_path_count(n, types, e, src, dst, on_path);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Attribute resync of the sampling CLI: AIGgan.py:29-62 (resync_attrs).
 * inv_cnt[v] = number of entering edges with inversion bit == 1 (not clamped);
 * depth[v] = longest-path level over in-range edges, all zeros if the graph has a cycle.
 * ------------------------------------------------------------------------------------------ */
int orc_resync(int64_t n, int64_t e, const int64_t *src, const int64_t *dst, const int64_t *inv,
               int64_t *inv_cnt, int64_t *depth) {
    for (int64_t i = 0; i < n; ++i) { inv_cnt[i] = 0; depth[i] = 0; }
    for (int64_t i = 0; i < e; ++i) if (inv[i] == 1) inv_cnt[dst[i]]++;
    int64_t *fs = (int64_t *)malloc(sizeof(int64_t) * (size_t)(e > 0 ? e : 1) * 2), *fd = fs + e, m = 0;
    for (int64_t i = 0; i < e; ++i)
        if (src[i] >= 0 && src[i] < n && dst[i] >= 0 && dst[i] < n) { fs[m] = src[i]; fd[m] = dst[i]; ++m; }
    int64_t *ptr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1));
    int64_t *adj = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m > 0 ? m : 1));
    int64_t *deg = (int64_t *)calloc((size_t)(n > 0 ? n : 1), sizeof(int64_t));
    int64_t *q = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
    build_csr(n, m, fs, fd, ptr, adj);
    for (int64_t i = 0; i < m; ++i) deg[fd[i]]++;
    int64_t head = 0, tail = 0;
    for (int64_t i = 0; i < n; ++i) if (!deg[i]) q[tail++] = i;
    while (head < tail) {
        int64_t u = q[head++];
        for (int64_t k = ptr[u]; k < ptr[u + 1]; ++k) {
            int64_t v = adj[k];
            if (depth[u] + 1 > depth[v]) depth[v] = depth[u] + 1;
            if (--deg[v] == 0) q[tail++] = v;
        }
    }
    if (tail != n) for (int64_t i = 0; i < n; ++i) depth[i] = 0;
    free(fs); free(ptr); free(adj); free(deg); free(q);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Floating-point half of the generator in a position-independent form.
 *
 * models/generator.py:115-117 (node_enc, node_proj) and :94-98 / :169 (edge_scorer,
 * edge_attr_pred) are dense layers.  The reference evaluates them through the CPU BLAS, whose
 * blocked kernels round the remainder rows of a matrix differently from the full blocks: equal
 * input rows can come out 1-3 ulp apart depending on where they sit (observed: 59 identical
 * candidate rows gave 3 distinct logits).  Its pick among tied candidates is therefore not a
 * function of the inputs alone.  The library defines the arithmetic instead: every output element
 * is one fmaf chain over k in a fixed order, so equal rows give equal results anywhere.  These
 * routines state that chain on the CPU; results agree with the reference's to ~1e-6 relative.
 *
 * Chain order of a 128-wide product: k runs 8g+0, 8g+4, 8g+1, 8g+5, 8g+2, 8g+6, 8g+3, 8g+7 for
 * g = 0..15 (the order in which the matrix-core kernel feeds k), accumulator starts at 0, the
 * bias is added last.  Compile with -ffp-contract=off: every fused operation below is explicit.
 * ------------------------------------------------------------------------------------------ */
#include <math.h>

static float chain128(const float *a, const float *w) {
    float acc = 0.0f;
    for (int g = 0; g < 16; ++g)
        for (int t = 0; t < 4; ++t) {
            acc = fmaf(a[8 * g + t], w[8 * g + t], acc);
            acc = fmaf(a[8 * g + 4 + t], w[8 * g + 4 + t], acc);
        }
    return acc;
}

/* out[j] = b[j] + sum_k w[j*ld + off + k] * z[k], k < zd ascending */
static void zbias128(const float *w, int ld, int off, const float *b, const float *z, int zd, float *out) {
    for (int j = 0; j < 128; ++j) {
        float acc = 0.0f;
        for (int k = 0; k < zd; ++k) acc = fmaf(w[(size_t)j * ld + off + k], z[k], acc);
        out[j] = acc + b[j];
    }
}

static void layer128(int64_t n, const float *x, const float *w, int ldw, const float *bias, int relu, float *y) {
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 128; ++j) {
            float v = chain128(x + i * 128, w + (size_t)j * ldw) + (bias ? bias[j] : 0.0f);
            y[i * 128 + j] = (relu && !(v > 0.0f)) ? 0.0f : v;
        }
}

/* weights: the 16 tensors of AIGGenerator in state_dict order, PyTorch [out,in] layout.
 * h_cond [n,128]; proj = 4 planes [n,128]: source half / target half(+z,+bias) of edge_scorer.0,
 * then the same two for edge_attr_pred.0. */
int orc_node_proj_z(int64_t n, const float *x, const float *z, int zd,
                    const float *enc0_w, const float *enc0_b, const float *enc2_w, const float *enc2_b,
                    const float *proj0_w, const float *proj0_b, const float *proj2_w, const float *proj2_b,
                    const float *sc0_w, const float *sc0_b, const float *at0_w, const float *at0_b,
                    float *h_cond, float *proj);

int orc_node_proj(int64_t n, const float *x, const float *z,
                  const float *enc0_w, const float *enc0_b, const float *enc2_w, const float *enc2_b,
                  const float *proj0_w, const float *proj0_b, const float *proj2_w, const float *proj2_b,
                  const float *sc0_w, const float *sc0_b, const float *at0_w, const float *at0_b,
                  float *h_cond, float *proj) {
    return orc_node_proj_z(n, x, z, 128, enc0_w, enc0_b, enc2_w, enc2_b, proj0_w, proj0_b, proj2_w, proj2_b, sc0_w, sc0_b,
                           at0_w, at0_b, h_cond, proj);
}

/* the same for a z of zd values (--z-dim of train.py:374 / AIGgan.py:78; models/generator.py:24-41: node_proj.0 is
 * Linear(128 + zd, 128), the edge MLPs' first layers Linear(256 + zd, 128)) */
int orc_node_proj_z(int64_t n, const float *x, const float *z, int zd,
                    const float *enc0_w, const float *enc0_b, const float *enc2_w, const float *enc2_b,
                    const float *proj0_w, const float *proj0_b, const float *proj2_w, const float *proj2_b,
                    const float *sc0_w, const float *sc0_b, const float *at0_w, const float *at0_b,
                    float *h_cond, float *proj) {
    const int ldp = 128 + zd, lde = 256 + zd;
    float zb[3][128];
    zbias128(proj0_w, ldp, 128, proj0_b, z, zd, zb[0]);
    zbias128(sc0_w, lde, 256, sc0_b, z, zd, zb[1]);
    zbias128(at0_w, lde, 256, at0_b, z, zd, zb[2]);
    float *t0 = (float *)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1) * 128 * 2), *t1 = t0 + n * 128;
    if (!t0) return -1;
    for (int64_t i = 0; i < n; ++i)
        for (int j = 0; j < 128; ++j) {
            float s = fmaf(x[2 * i + 1], enc0_w[2 * j + 1], x[2 * i] * enc0_w[2 * j]) + enc0_b[j];
            t0[i * 128 + j] = s > 0.0f ? s : 0.0f;
        }
    layer128(n, t0, enc2_w, 128, enc2_b, 1, t1);
    layer128(n, t1, proj0_w, ldp, zb[0], 1, t0);
    layer128(n, t0, proj2_w, 128, proj2_b, 1, h_cond);
    layer128(n, h_cond, sc0_w, lde, NULL, 0, proj);
    layer128(n, h_cond, sc0_w + 128, lde, zb[1], 0, proj + n * 128);
    layer128(n, h_cond, at0_w, lde, NULL, 0, proj + 2 * n * 128);
    layer128(n, h_cond, at0_w + 128, lde, zb[2], 0, proj + 3 * n * 128);
    free(t0);
    return 0;
}

/* logit of the pair (source row a, target row b): b2 + sum_j w2[j] * relu(a[j] + b[j]), j ascending */
float orc_pair_logit(const float *a, const float *b, const float *w2, float b2) {
    float acc = 0.0f;
    for (int j = 0; j < 128; ++j) {
        float h = a[j] + b[j];
        acc = fmaf(h > 0.0f ? h : 0.0f, w2[j], acc);
    }
    return acc + b2;
}

/* logits[t*K + i] for every target t (processing order) and candidate i < cnt[t]; others untouched */
int orc_score_rows(int64_t T, int64_t K, const int64_t *tgt, const int64_t *cnt, const int64_t *cand,
                   const float *A, const float *B, const float *w2, const float *b2, float *logits) {
    for (int64_t t = 0; t < T; ++t)
        for (int64_t i = 0; i < cnt[t]; ++i)
            logits[t * K + i] = orc_pair_logit(A + cand[t * K + i] * 128, B + tgt[t] * 128, w2, b2[0]);
    return 0;
}

/* inversion bits of given (src, dst) pairs: (w2a . relu(A2[src] + B2[dst]) + b2a) > 0 */
int orc_inv_bits(int64_t m, const int64_t *src, const int64_t *dst, const float *A2, const float *B2,
                 const float *w2a, const float *b2a, int64_t *bits) {
    for (int64_t i = 0; i < m; ++i)
        bits[i] = orc_pair_logit(A2 + src[i] * 128, B2 + dst[i] * 128, w2a, b2a[0]) > 0.0f;
    return 0;
}
