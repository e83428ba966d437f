// Native writers of the slide-level outputs (reference: json.dump of per-cell dicts,
// cell_segmentation/inference/cell_detection.py:438-457).  The reference serialises ~10^6 Python dicts per slide through the
// pure-Python JSON encoder (indent=2): tens of seconds and gigabytes per slide — with the tile loop on 8 GPUs that is the whole
// wall-clock.  Here the files are rendered from the packed record arrays by host code in this library (no Python object per
// cell; called through ctypes, so the GIL is released and cells.pt is pickled concurrently).  Same JSON documents: same keys
// in the same order, same values (doubles with 17 significant digits: they parse to the identical binary64), one cell per line.
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <charconv>
#include <string>
#include <thread>
#include <vector>

#include "../../include/cellvit_amd.h"

void cva_set_error(const char* fmt, ...);

namespace {

// One rendering buffer.  A slide's cells.json is ~0.8 KB per cell, ~80 integers each: the first version went through snprintf per number and one thread —
// 80 MB/s, 3.2 s of a 14-s slide (profiles/r04_zz_slide_*).  Integers are now written by hand, doubles by std::to_chars(general, 17) — byte for byte what
// "%.17g" prints, five times faster —, and chunks of cells are rendered by several threads into their own buffers and written in order.
struct Out {
    std::vector<char> buf;
    void put(const char* s, size_t n) { buf.insert(buf.end(), s, s + n); }
    void put(const char* s) { put(s, strlen(s)); }
    void i64(long long v) {
        char t[24]; int n = 24;
        unsigned long long u = v < 0 ? 0ull - (unsigned long long)v : (unsigned long long)v;
        do { t[--n] = (char)('0' + u % 10); u /= 10; } while (u);
        if (v < 0) t[--n] = '-';
        put(t + n, 24 - n);
    }
    void f64(double v) {
        char t[40];
        // non-finite values as Python's json encoder writes them (NaN / Infinity / -Infinity: what json.load parses back)
        if (v != v) { put("NaN"); return; }
        if (v - v != 0.0) { put(v > 0 ? "Infinity" : "-Infinity"); return; }
        int n = (int)(std::to_chars(t, t + sizeof t, v, std::chars_format::general, 17).ptr - t);      // == snprintf("%.17g")
        // JSON numbers as Python writes floats: always with a fraction or exponent ("3.0", not "3")
        bool plain = true;
        for (int k = 0; k < n; ++k) if (t[k] == '.' || t[k] == 'e') { plain = false; break; }
        if (plain) { t[n++] = '.'; t[n++] = '0'; }
        put(t, n);
    }
};

// get_edge_patch (cell_detection.py:877-902): [top, right, down, left] -> neighbour tile offsets (drow, dcol), or none
const int8_t EDGE_TABLE[16][7] = {   // n, then up to 3 (dr, dc) pairs; index = top*8 + right*4 + down*2 + left
    {-1}, {1, 0, -1}, {1, 1, 0}, {3, 1, 0, 1, -1, 0, -1}, {1, 0, 1}, {-1}, {3, 0, 1, 1, 1, 1, 0}, {-1},
    {1, -1, 0}, {3, 0, -1, -1, -1, -1, 0}, {-1}, {-1}, {3, -1, 0, -1, 1, 0, 1}, {-1}, {-1}, {-1}};

// One round of chunk rendering: chunks [c0, c0 + nc) on up to nc threads, each into its own buffer.  Nothing may throw through the
// extern "C" entry points (ctypes callers would be terminated): thread exhaustion (std::system_error) falls back to rendering the
// remaining chunks on the calling thread, every started thread is joined, and allocation failures inside a chunk (std::bad_alloc from
// the buffers) are caught per chunk and reported.
template <class Render>
bool render_round(int nc, std::vector<Out>& outs, Render&& render_chunk) {
    std::vector<std::thread> th;
    std::vector<char> failed((size_t)nc, 0);
    auto guarded = [&](int t) {
        try { outs[t].buf.clear(); render_chunk(t, outs[t]); } catch (...) { failed[t] = 1; }
    };
    int started = 1;                                   // chunk 0 runs on the calling thread
    try {
        th.reserve((size_t)nc);
        for (int t = 1; t < nc; ++t) { th.emplace_back(guarded, t); started = t + 1; }
    } catch (...) { /* no more threads: the rest is rendered here */ }
    guarded(0);
    for (int t = started; t < nc; ++t) guarded(t);
    for (auto& x : th) x.join();
    for (int t = 0; t < nc; ++t) if (failed[t]) return false;
    return true;
}

}  // namespace

extern "C" int cv_write_cells_json(const char* path, const char* header, int detection_only, int n, const int64_t* bbox,
                                   const double* centroid, const int64_t* ct_off, const int64_t* ct_xy, const double* type_prob,
                                   const int32_t* type, const int32_t* patch_rc, const int32_t* status, const int64_t* offset_global,
                                   const uint8_t* edge, const uint8_t* edge_pos) {
    if (!path || !header || n < 0 || (n && (!bbox || !centroid || !type))) { cva_set_error("cv_write_cells_json: bad argument"); return CV_ERR_INVALID; }
    if (!detection_only && n && (!ct_off || !ct_xy || !type_prob || !patch_rc || !status || !offset_global || !edge || !edge_pos)) {
        cva_set_error("cv_write_cells_json: bad argument"); return CV_ERR_INVALID;
    }
    FILE* f = fopen(path, "wb");
    if (!f) { cva_set_error("cv_write_cells_json: cannot open %s", path); return CV_ERR_INVALID; }
    bool wrote = true;
    auto emit = [&](const std::vector<char>& b) { if (!b.empty() && fwrite(b.data(), 1, b.size(), f) != b.size()) wrote = false; };
    {
        Out o;
        o.put("{");
        o.put(header);                               // '"wsi_metadata": {...}, "processed_patches": [...], "type_map": {...}' rendered by the caller
        o.put(", \"cells\": [");
        emit(o.buf);
    }
    auto render = [&](int k0, int k1, Out& o) {
        for (int k = k0; k < k1; ++k) {
            o.put(k ? ",\n{\"bbox\": [[" : "\n{\"bbox\": [[");
            o.i64(bbox[4 * k]); o.put(", "); o.i64(bbox[4 * k + 1]); o.put("], ["); o.i64(bbox[4 * k + 2]); o.put(", "); o.i64(bbox[4 * k + 3]);
            o.put("]], \"centroid\": ["); o.f64(centroid[2 * k]); o.put(", "); o.f64(centroid[2 * k + 1]); o.put("]");
            if (detection_only) {
                o.put(", \"type\": "); o.i64(type[k]); o.put("}");
                continue;
            }
            o.put(", \"contour\": [");
            for (int64_t q = ct_off[k]; q < ct_off[k + 1]; ++q) {
                o.put(q == ct_off[k] ? "[" : ", ["); o.i64(ct_xy[2 * q]); o.put(", "); o.i64(ct_xy[2 * q + 1]); o.put("]");
            }
            o.put("], \"type_prob\": "); o.f64(type_prob[k]);
            o.put(", \"type\": "); o.i64(type[k]);
            o.put(", \"patch_coordinates\": ["); o.i64(patch_rc[2 * k]); o.put(", "); o.i64(patch_rc[2 * k + 1]);
            o.put("], \"cell_status\": "); o.i64(status[k]);
            o.put(", \"offset_global\": ["); o.i64(offset_global[2 * k]); o.put(", "); o.i64(offset_global[2 * k + 1]); o.put("]");
            if (edge[k]) {
                const uint8_t* ps = edge_pos + 4 * k;
                o.put(", \"edge_position\": true, \"edge_information\": {\"position\": [");
                o.i64(ps[0]); o.put(", "); o.i64(ps[1]); o.put(", "); o.i64(ps[2]); o.put(", "); o.i64(ps[3]);
                o.put("], \"edge_patches\": ");
                const int8_t* e = EDGE_TABLE[(ps[0] & 1) * 8 + (ps[1] & 1) * 4 + (ps[2] & 1) * 2 + (ps[3] & 1)];
                if (e[0] < 0) o.put("null");
                else {
                    o.put("[");
                    for (int q = 0; q < e[0]; ++q) {
                        o.put(q ? ", [" : "["); o.i64(patch_rc[2 * k] + e[1 + 2 * q]); o.put(", "); o.i64(patch_rc[2 * k + 1] + e[2 + 2 * q]); o.put("]");
                    }
                    o.put("]");
                }
                o.put("}}");
            } else {
                o.put(", \"edge_position\": false}");
            }
        }
    };
    // chunks of cells rendered concurrently (each into its own buffer), written in order; a round holds at most `nthr` chunks in memory
    constexpr int CHUNK = 8192;
    const int nchunks = (n + CHUNK - 1) / CHUNK;
    const int nthr = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, nchunks}));
    bool rendered = true;
    try {
        std::vector<Out> outs(nthr);
        for (int c0 = 0; c0 < nchunks && wrote && rendered; c0 += nthr) {
            const int nc = std::min(nthr, nchunks - c0);
            rendered = render_round(nc, outs, [&](int t, Out& o) { render((c0 + t) * CHUNK, std::min(n, (c0 + t + 1) * CHUNK), o); });
            for (int t = 0; t < nc && rendered; ++t) emit(outs[t].buf);
        }
        Out o;
        o.put(n ? "\n]}" : "]}");
        emit(o.buf);
    } catch (...) { rendered = false; }
    const bool closed = fclose(f) == 0;          // always closed, also after a failed write / a failed allocation
    const bool ok = wrote && closed && rendered;
    if (!ok) { cva_set_error("cv_write_cells_json: write to %s failed", path); return CV_ERR_INVALID; }
    return CV_OK;
}


// The optional geojson pair (cell_detection.py:538-597 + template_geojson.py:9-52): one Feature per nucleus type present, in ascending type order
// (`sorted(df.type.unique())`), whose geometry collects every cell of that type in slide order — MultiPolygon of the closed contour rings (cells.geojson) or
// MultiPoint of the centroids (cell_detection.geojson).  The reference builds 10^6 Python lists and json.dump's them with indent=2; here the features' heads
// and tails (type, uuid, properties: rendered by the caller) frame coordinates rendered from the arrays, chunks of cells on several threads as above.
// Same document after parsing (the reference's file differs in whitespace and in its random ids anyway).
extern "C" int cv_write_geojson(const char* path, int polygons, int n, const double* centroid, const int64_t* ct_off, const int64_t* ct_xy,
                                const int32_t* type, int n_feat, const int32_t* feat_type, const char* const* feat_head, const char* const* feat_tail) {
    if (!path || n < 0 || n_feat < 0 || (n && (!type || (polygons ? (!ct_off || !ct_xy) : !centroid))) || (n_feat && (!feat_type || !feat_head || !feat_tail))) {
        cva_set_error("cv_write_geojson: bad argument"); return CV_ERR_INVALID;
    }
    FILE* f = fopen(path, "wb");
    if (!f) { cva_set_error("cv_write_geojson: cannot open %s", path); return CV_ERR_INVALID; }
    bool wrote = true, rendered = true;
    auto emit = [&](const char* b, size_t len) { if (len && fwrite(b, 1, len, f) != len) wrote = false; };
    emit("[", 1);
    constexpr int CHUNK = 8192;
    try {
    std::vector<int> cells;
    for (int ft = 0; ft < n_feat && wrote && rendered; ++ft) {
        cells.clear();
        for (int k = 0; k < n; ++k) if (type[k] == feat_type[ft]) cells.push_back(k);
        if (ft) emit(", ", 2);
        emit(feat_head[ft], strlen(feat_head[ft]));
        const int m = (int)cells.size();
        auto render = [&](int a, int b, Out& o) {
            for (int c = a; c < b; ++c) {
                const int k = cells[c];
                if (c) o.put(", ");
                if (!polygons) { o.put("["); o.f64(centroid[2 * k]); o.put(", "); o.f64(centroid[2 * k + 1]); o.put("]"); continue; }
                o.put("[[");
                for (int64_t q = ct_off[k]; q < ct_off[k + 1]; ++q) {            // the integer contour lists as they are (cell_detection.py:358-363, 564-568)
                    o.put(q == ct_off[k] ? "[" : ", ["); o.i64(ct_xy[2 * q]); o.put(", "); o.i64(ct_xy[2 * q + 1]); o.put("]");
                }
                if (ct_off[k + 1] > ct_off[k]) {                                 // c.append(c[0])
                    const int64_t q = ct_off[k];
                    o.put(", ["); o.i64(ct_xy[2 * q]); o.put(", "); o.i64(ct_xy[2 * q + 1]); o.put("]");
                }
                o.put("]]");
            }
        };
        const int nchunks = (m + CHUNK - 1) / CHUNK;
        const int nthr = std::max(1, std::min({(int)std::thread::hardware_concurrency(), 16, nchunks}));
        std::vector<Out> outs(nthr);
        for (int c0 = 0; c0 < nchunks && wrote && rendered; c0 += nthr) {
            const int nc = std::min(nthr, nchunks - c0);
            rendered = render_round(nc, outs, [&](int t, Out& o) { render((c0 + t) * CHUNK, std::min(m, (c0 + t + 1) * CHUNK), o); });
            for (int t = 0; t < nc && rendered; ++t) emit(outs[t].buf.data(), outs[t].buf.size());
        }
        emit(feat_tail[ft], strlen(feat_tail[ft]));
    }
    } catch (...) { rendered = false; }
    emit("]", 1);
    const bool closed = fclose(f) == 0;
    if (!(wrote && closed && rendered)) { cva_set_error("cv_write_geojson: write to %s failed", path); return CV_ERR_INVALID; }
    return CV_OK;
}
