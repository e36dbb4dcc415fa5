// csr_build.cpp - one-time host build of the two CSR orientations of a batched mesh graph.
//
// Replaces the per-call gather/scatter index handling of PyG's MessagePassing.propagate
// (/root/reference/src/GRAND_plus.py:233-234) - see include/gadapt_hip.h.  Stable counting
// sorts: edges keep their input order inside a row, so the summation order of every
// kernel is a pure function of the caller's edge list.
#include <stdint.h>
#include <stdio.h>
#include <vector>
#include "gadapt_hip.h"

extern "C" int gadapt_csr_build_host(const int64_t* src, const int64_t* dst, int64_t n_edges, int64_t n_nodes,
                                     int32_t* rowptr_t, int32_t* col_t, int32_t* eid_t,
                                     int32_t* rowptr_s, int32_t* col_s, int32_t* perm_s, int32_t* tpos_s) {
    if (!src || !dst || n_edges < 0 || n_nodes <= 0 || n_nodes > INT32_MAX || n_edges > INT32_MAX ||
        !rowptr_t || !col_t || !eid_t || !rowptr_s || !col_s || !perm_s || !tpos_s)
        return GADAPT_E_BADARG;
    const int64_t N = n_nodes, E = n_edges;
    for (int64_t i = 0; i <= N; ++i) rowptr_t[i] = rowptr_s[i] = 0;
    for (int64_t e = 0; e < E; ++e) {
        if (src[e] < 0 || src[e] >= N || dst[e] < 0 || dst[e] >= N) return GADAPT_E_RANGE;
        rowptr_t[dst[e] + 1]++;
        rowptr_s[src[e] + 1]++;
    }
    for (int64_t i = 0; i < N; ++i) { rowptr_t[i + 1] += rowptr_t[i]; rowptr_s[i + 1] += rowptr_s[i]; }
    std::vector<int32_t> fill_t(rowptr_t, rowptr_t + N), fill_s(rowptr_s, rowptr_s + N);
    // by target, stable in input order
    for (int64_t e = 0; e < E; ++e) {
        const int32_t slot = fill_t[dst[e]]++;
        col_t[slot] = (int32_t)src[e];
        eid_t[slot] = (int32_t)e;
    }
    // by source, stable in TARGET-SLOT order (so perm_s is monotone inside a row where possible)
    for (int32_t slot = 0; slot < (int32_t)E; ++slot) {
        const int64_t e = eid_t[slot];
        const int32_t s = fill_s[src[e]]++;
        col_s[s] = (int32_t)dst[e];
        perm_s[s] = slot;
        tpos_s[slot] = s;
    }
    return GADAPT_OK;
}

// Per-tile staging metadata (see include/gadapt_hip.h): 4 ints per tile of `tile_rows` consecutive rows.
extern "C" int gadapt_tile_meta_host(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int tile_rows, int32_t* meta_out) {
    if (!rowptr || !col || !meta_out || n_nodes <= 0 || tile_rows <= 0) return GADAPT_E_BADARG;
    const int64_t n_tiles = (n_nodes + tile_rows - 1) / tile_rows;
    for (int64_t t = 0; t < n_tiles; ++t) {
        const int64_t lo = t * tile_rows, hi = (lo + tile_rows < n_nodes) ? lo + tile_rows : n_nodes;
        int32_t longest = 0;
        for (int64_t i = lo; i < hi; ++i) { const int32_t d = rowptr[i + 1] - rowptr[i]; if (d > longest) longest = d; }
        // windowed: every neighbour of the tile's rows lies in the slabs t-1, t, t+1
        const int64_t wlo = (t - 1) * tile_rows, whi = (t + 2) * tile_rows;
        int32_t windowed = 1;
        for (int32_t e = rowptr[lo]; e < rowptr[hi]; ++e)
            if (col[e] < wlo || col[e] >= whi) { windowed = 0; break; }
        meta_out[4 * t + 0] = rowptr[lo];
        meta_out[4 * t + 1] = rowptr[hi] - rowptr[lo];
        meta_out[4 * t + 2] = longest;
        meta_out[4 * t + 3] = windowed;
    }
    return GADAPT_OK;
}

// ELL-8 copy of one CSR orientation for the wide kernels (see include/gadapt_hip.h): row i -> ell[8i..8i+7], unused
// entries -1, rows padded to a multiple of 256 (rows longer than 8 are cut: such a graph never qualifies).  *max_deg_out = the
// longest row if the orientation qualifies for the 384-row window (every row <= 8 entries and every neighbour of node i
// inside rows [256*(i/256) - 64, 256*(i/256) + 320)), else 0.
extern "C" int gadapt_ell_build_host(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int32_t* ell_out, int32_t* max_deg_out) {
    if (!rowptr || !col || !ell_out || !max_deg_out || n_nodes <= 0) return GADAPT_E_BADARG;
    const int64_t n_pad = (n_nodes + 255) / 256 * 256;
    for (int64_t k = 0; k < n_pad * 8; ++k) ell_out[k] = -1;
    for (int64_t i = 0; i < n_nodes; ++i) {
        const int32_t e0 = rowptr[i], d = rowptr[i + 1] - e0;
        for (int32_t k = 0; k < d && k < 8; ++k) ell_out[8 * i + k] = col[e0 + k];
    }
    return gadapt_wide_window_host(rowptr, col, n_nodes, 256, 64, 8, max_deg_out);
}

// The wide kernels' locality test for steps of `step` nodes and a window of step + 2 * halo rows: *max_deg_out = the longest row if every
// row has at most max_row entries and every neighbour of node i lies in rows [step*(i/step) - halo, step*(i/step) + step + halo), else 0.
extern "C" int gadapt_wide_window_host(const int32_t* rowptr, const int32_t* col, int64_t n_nodes, int step, int halo, int max_row, int32_t* max_deg_out) {
    if (!rowptr || !col || !max_deg_out || n_nodes <= 0 || step <= 0 || halo < 0 || max_row <= 0) return GADAPT_E_BADARG;
    int32_t longest = 0;
    bool ok = true;
    for (int64_t i = 0; i < n_nodes && ok; ++i) {
        const int32_t e0 = rowptr[i], d = rowptr[i + 1] - e0;
        if (d > max_row) { ok = false; break; }
        if (d > longest) longest = d;
        const int64_t lo = i / step * step - halo, hi = lo + step + 2 * (int64_t)halo;
        for (int32_t k = 0; k < d; ++k) {
            const int32_t j = col[e0 + k];
            if (j < lo || j >= hi) { ok = false; break; }
        }
    }
    *max_deg_out = ok ? longest : 0;
    return GADAPT_OK;
}
