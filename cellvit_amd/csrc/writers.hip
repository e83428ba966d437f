// Native writers of the slide-level outputs (reference: json.dump of per-cell dicts,
// cell_segmentation/inference/cell_detection.py:438-457).  The reference serialises ~10^6 Python dicts per slide through the
// pure-Python JSON encoder (indent=2): tens of seconds and gigabytes per slide — with the tile loop on 8 GPUs that is the whole
// wall-clock.  Here the files are rendered from the packed record arrays by host code in this library (no Python object per
// cell; called through ctypes, so the GIL is released and cells.pt is pickled concurrently).  Same JSON documents: same keys
// in the same order, same values (doubles with 17 significant digits: they parse to the identical binary64), one cell per line.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <unistd.h>

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

// One cell of cells.json / cell_detection.json: "bbox": ... (the text between the opening brace's `{"bbox": [[` and the cell's closing brace included)
struct CellArrays {
    int detection_only;
    const int64_t* bbox; const double* centroid; const int64_t* ct_off; const int64_t* ct_xy; const double* type_prob; const int32_t* type;
    const int32_t* patch_rc; const int32_t* status; const int64_t* offset_global; const uint8_t* edge; const uint8_t* edge_pos;
};
void render_cell_tail(Out& o, const CellArrays& a, int k) {
    const int detection_only = a.detection_only;
    const int64_t* bbox = a.bbox; const double* centroid = a.centroid; const int64_t* ct_off = a.ct_off; const int64_t* ct_xy = a.ct_xy;
    const double* type_prob = a.type_prob; const int32_t* type = a.type; const int32_t* patch_rc = a.patch_rc; const int32_t* status = a.status;
    const int64_t* offset_global = a.offset_global; const uint8_t* edge = a.edge; const uint8_t* edge_pos = a.edge_pos;

            o.i64(bbox[4 * k]); o.put(", "); o.i64(bbox[4 * k + 1]); o.put("], ["); o.i64(bbox[4 * k + 2]); o.put(", "); o.i64(bbox[4 * k + 3]);
            o.put("]], \"centroid\": ["); o.f64(centroid[2 * k]); o.put(", "); o.f64(centroid[2 * k + 1]); o.put("]");
            if (detection_only) {
                o.put(", \"type\": "); o.i64(type[k]); o.put("}");
                return;
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
    const CellArrays ca{detection_only, bbox, centroid, ct_off, ct_xy, type_prob, type, patch_rc, status, offset_global, edge, edge_pos};
    auto render = [&](int k0, int k1, Out& o) {
        for (int k = k0; k < k1; ++k) {
            o.put(k ? ",\n{\"bbox\": [[" : "\n{\"bbox\": [[");
            render_cell_tail(o, ca, k);
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


// ------------------------------------------------------------------------------------------------------------------------------------
// Streaming slide tail (round 5; reference: the writers run after the whole tile loop, cell_detection.py:423-475).  While the tile loop
// runs, every finished batch's cells are rendered ONCE into a text buffer (one JSON object per cell, with offsets); after the slide-level
// de-duplication the kept cells of a batch are joined by one pass over that buffer (cv_textbuf_compact) and the chunks are written in
// slide order — the files are byte for byte what cv_write_cells_json writes for the kept cells.  cells.pt: the archive is written with
// torch.serialization.skip_data (headers + pickle, holes for the tensor bytes) and the holes are filled by cv_write_rows: kept rows gathered
// from the host arenas the batches' token rows were copied to during the loop, written with pwrite on several threads, CRC-32 returned for
// the zip headers.
// ------------------------------------------------------------------------------------------------------------------------------------
struct cv_textbuf { std::vector<char> text; std::vector<int64_t> off; };

extern "C" int cv_render_cells(int detection_only, int n, const int64_t* bbox, const double* centroid, const int64_t* ct_off, const int64_t* ct_xy,
                               const double* type_prob, const int32_t* type, const int32_t* patch_rc, const int32_t* status,
                               const int64_t* offset_global, const uint8_t* edge, const uint8_t* edge_pos, cv_textbuf** out) {
    if (!out || n < 0 || (n && (!bbox || !centroid || !type))) { cva_set_error("cv_render_cells: bad argument"); return CV_ERR_INVALID; }
    if (!detection_only && n && (!ct_off || !ct_xy || !type_prob || !patch_rc || !status || !offset_global || !edge || !edge_pos)) {
        cva_set_error("cv_render_cells: bad argument"); return CV_ERR_INVALID;
    }
    *out = nullptr;
    try {
        cv_textbuf* b = new cv_textbuf();
        b->off.resize((size_t)n + 1);
        const CellArrays ca{detection_only, bbox, centroid, ct_off, ct_xy, type_prob, type, patch_rc, status, offset_global, edge, edge_pos};
        Out o;
        o.buf.reserve((size_t)n * (detection_only ? 96 : 800));
        for (int k = 0; k < n; ++k) {
            b->off[k] = (int64_t)o.buf.size();
            o.put("{\"bbox\": [[");
            render_cell_tail(o, ca, k);
        }
        b->off[n] = (int64_t)o.buf.size();
        b->text.swap(o.buf);
        *out = b;
    } catch (...) { cva_set_error("cv_render_cells: out of memory"); return CV_ERR_INVALID; }
    return CV_OK;
}

// Bytes of the kept cells (keep[k] != 0; keep == NULL: all) joined by ",\n".  dst == NULL or cap too small: nothing is written.  Returns the size.
extern "C" int64_t cv_textbuf_compact(const cv_textbuf* b, const uint8_t* keep, char* dst, int64_t cap) {
    if (!b) return -1;
    const int n = (int)b->off.size() - 1;
    int64_t need = 0; int cnt = 0;
    for (int k = 0; k < n; ++k) if (!keep || keep[k]) { need += b->off[k + 1] - b->off[k]; ++cnt; }
    if (cnt > 1) need += 2 * (int64_t)(cnt - 1);
    if (!dst || cap < need) return need;
    char* w = dst; bool first = true;
    for (int k = 0; k < n; ++k) {
        if (keep && !keep[k]) continue;
        if (!first) { *w++ = ','; *w++ = '\n'; }
        first = false;
        const int64_t len = b->off[k + 1] - b->off[k];
        memcpy(w, b->text.data() + b->off[k], (size_t)len);
        w += len;
    }
    return need;
}

extern "C" void cv_textbuf_free(cv_textbuf* b) { delete b; }

namespace {
// CRC-32 (IEEE 802.3, the zip polynomial), slicing by 8; combination of the CRCs of adjacent blocks as in zlib's crc32_combine (the
// operator "append len2 zero bytes" as a GF(2) matrix, applied by repeated squaring)
struct Crc32 {
    uint32_t t[8][256];
    Crc32() {
        for (uint32_t i = 0; i < 256; ++i) { uint32_t c = i; for (int k = 0; k < 8; ++k) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t[0][i] = c; }
        for (uint32_t i = 0; i < 256; ++i) for (int s = 1; s < 8; ++s) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xff];
    }
    uint32_t run(uint32_t crc, const unsigned char* p, size_t n) const {
        crc = ~crc;
        while (n && ((uintptr_t)p & 7)) { crc = t[0][(crc ^ *p++) & 0xff] ^ (crc >> 8); --n; }
        while (n >= 8) {
            uint64_t v; memcpy(&v, p, 8);
            const uint32_t lo = (uint32_t)v ^ crc, hi = (uint32_t)(v >> 32);
            crc = t[7][lo & 0xff] ^ t[6][(lo >> 8) & 0xff] ^ t[5][(lo >> 16) & 0xff] ^ t[4][lo >> 24] ^
                  t[3][hi & 0xff] ^ t[2][(hi >> 8) & 0xff] ^ t[1][(hi >> 16) & 0xff] ^ t[0][hi >> 24];
            p += 8; n -= 8;
        }
        while (n--) crc = t[0][(crc ^ *p++) & 0xff] ^ (crc >> 8);
        return ~crc;
    }
};
const Crc32& crc_tab() { static const Crc32 c; return c; }
uint32_t gf2_times(const uint32_t* mat, uint32_t vec) { uint32_t s = 0; for (; vec; vec >>= 1, ++mat) if (vec & 1) s ^= *mat; return s; }
void gf2_square(uint32_t* sq, const uint32_t* mat) { for (int n = 0; n < 32; ++n) sq[n] = gf2_times(mat, mat[n]); }
uint32_t crc32_combine(uint32_t crc1, uint32_t crc2, uint64_t len2) {
    if (!len2) return crc1;
    uint32_t even[32], odd[32];
    odd[0] = 0xEDB88320u;
    uint32_t row = 1;
    for (int n = 1; n < 32; ++n) { odd[n] = row; row <<= 1; }
    gf2_square(even, odd);
    gf2_square(odd, even);
    do {
        gf2_square(even, odd);
        if (len2 & 1) crc1 = gf2_times(even, crc1);
        len2 >>= 1;
        if (!len2) break;
        gf2_square(odd, even);
        if (len2 & 1) crc1 = gf2_times(odd, crc1);
        len2 >>= 1;
    } while (len2);
    return crc1 ^ crc2;
}
}  // namespace

// Rows of `row_bytes` bytes, kept ones of every chunk in order (keep[c] == NULL: all rows of chunk c), written at file_offset of an EXISTING
// file with pwrite on up to 16 threads (each thread owns a contiguous range of output rows, gathers it through a 4-MiB buffer); *crc_out
// (optional) = CRC-32 of the bytes written.  Returns CV_OK, or CV_ERR_INVALID with the message in cv_last_error.
extern "C" int cv_write_rows(const char* path, int64_t file_offset, int64_t row_bytes, int n_chunks, const void* const* chunk_ptr,
                             const int64_t* chunk_rows, const uint8_t* const* keep, uint32_t* crc_out, int64_t* rows_out) {
    if (!path || file_offset < 0 || row_bytes <= 0 || n_chunks < 0 || (n_chunks && (!chunk_ptr || !chunk_rows))) {
        cva_set_error("cv_write_rows: bad argument"); return CV_ERR_INVALID;
    }
    try {
        // the kept rows as (chunk, row) runs; prefix of output rows per chunk
        std::vector<int64_t> kept_before((size_t)n_chunks + 1, 0);
        for (int c = 0; c < n_chunks; ++c) {
            int64_t k = 0;
            const uint8_t* m = keep ? keep[c] : nullptr;
            if (!m) k = chunk_rows[c]; else for (int64_t r = 0; r < chunk_rows[c]; ++r) k += m[r] != 0;
            kept_before[c + 1] = kept_before[c] + k;
        }
        const int64_t total = kept_before[n_chunks];
        if (rows_out) *rows_out = total;
        if (crc_out) *crc_out = 0;
        if (total == 0) return CV_OK;
        struct Fd {      // closed on every exit, an allocation failure below included
            int v;
            ~Fd() { if (v >= 0) close(v); }
        } fdg{open(path, O_WRONLY)};
        const int fd = fdg.v;
        if (fd < 0) { cva_set_error("cv_write_rows: cannot open %s", path); return CV_ERR_INVALID; }
        const int nthr = (int)std::max<int64_t>(1, std::min<int64_t>({(int64_t)std::thread::hardware_concurrency(), 16, (total * row_bytes) / (8 << 20) + 1}));
        std::vector<uint32_t> crcs((size_t)nthr, 0);
        std::vector<int64_t> lens((size_t)nthr, 0);
        std::vector<char> bad((size_t)nthr, 0);
        auto work = [&](int t) {
            try {
                const int64_t r0 = total * t / nthr, r1 = total * (t + 1) / nthr;       // output rows of this thread
                const size_t cap_rows = (size_t)std::max<int64_t>(1, (4 << 20) / row_bytes);
                std::vector<unsigned char> buf(cap_rows * (size_t)row_bytes);
                size_t fill = 0; int64_t out_row = r0; uint32_t crc = 0;
                auto flush = [&]() {
                    if (!fill) return true;
                    const size_t nb = fill * (size_t)row_bytes;
                    size_t done = 0;
                    while (done < nb) {
                        const ssize_t w = pwrite(fd, buf.data() + done, nb - done, (off_t)(file_offset + out_row * row_bytes + (int64_t)done));
                        if (w <= 0) return false;
                        done += (size_t)w;
                    }
                    crc = crc_tab().run(crc, buf.data(), nb);
                    out_row += (int64_t)fill; fill = 0;
                    return true;
                };
                int c = (int)(std::upper_bound(kept_before.begin(), kept_before.end(), r0) - kept_before.begin()) - 1;
                int64_t o = kept_before[c];                  // output row of the first kept row of chunk c
                for (; c < n_chunks && o < r1; ++c) {
                    const uint8_t* m = keep ? keep[c] : nullptr;
                    const unsigned char* src = reinterpret_cast<const unsigned char*>(chunk_ptr[c]);
                    for (int64_t r = 0; r < chunk_rows[c] && o < r1; ++r) {
                        if (m && !m[r]) continue;
                        if (o >= r0) {
                            memcpy(buf.data() + fill * (size_t)row_bytes, src + r * row_bytes, (size_t)row_bytes);
                            if (++fill == cap_rows && !flush()) { bad[t] = 1; return; }
                        }
                        ++o;
                    }
                }
                if (!flush()) { bad[t] = 1; return; }
                crcs[t] = crc; lens[t] = (r1 - r0) * row_bytes;
            } catch (...) { bad[t] = 1; }
        };
        std::vector<std::thread> th;
        int started = 1;
        try { for (int t = 1; t < nthr; ++t) { th.emplace_back(work, t); started = t + 1; } } catch (...) {}
        work(0);
        for (int t = started; t < nthr; ++t) work(t);
        for (auto& x : th) x.join();
        for (int t = 0; t < nthr; ++t) if (bad[t]) { cva_set_error("cv_write_rows: write to %s failed", path); return CV_ERR_INVALID; }
        if (crc_out) {
            uint32_t crc = crcs[0];
            for (int t = 1; t < nthr; ++t) crc = crc32_combine(crc, crcs[t], (uint64_t)lens[t]);
            *crc_out = crc;
        }
    } catch (...) { cva_set_error("cv_write_rows: out of memory"); return CV_ERR_INVALID; }
    return CV_OK;
}
