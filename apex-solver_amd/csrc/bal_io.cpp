// bal_io.cpp -- host-side input path: the BAL text reader, the problem
// builder of bin/bundle_adjustment.rs and the reference's lexicographic column layout, in C++.
//
// Reference (file:line under the apex-solver tree):
//   BalLoader::load / parse_header / parse_observations / parse_cameras / parse_points
//                                   crates/apex-io/src/bal.rs:138-388
//   BalCamera::normalize_focal_length  :100-114  (non-positive or non-finite focal -> 500.0)
//   axis_angle_to_so3 + SE3 variable   bin/bundle_adjustment.rs:200-208, 232-257
//   column order = sorted variable names  src/optimizer/mod.rs:530-536
// No device code: these entry points work without a GPU.
#include "../../include/apexgpu.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

struct apexgpu_bal {
    int64_t n_cam = 0, n_pt = 0, n_obs = 0;
    std::vector<uint32_t> cam_idx, pt_idx;
    std::vector<double> obs_uv;
    std::vector<double> cam_raw;  // 9 per camera: rx ry rz tx ty tz f k1 k2 (focal normalised)
    std::vector<double> points;   // 3 per point
};

static thread_local std::string g_bal_err;
static int bal_fail(int code, const std::string& msg) { g_bal_err = msg; return code; }

namespace {
struct Line { const char* b; const char* e; size_t no; };

// "lines().enumerate().map(trim).filter(!empty)" of bal.rs:145-150
struct LineIter {
    const char* p; const char* end; size_t no = 0;
    bool next(Line& out) {
        while (p < end) {
            const char* s = p;
            const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
            const char* le = nl ? nl : end;
            p = nl ? nl + 1 : end;
            ++no;
            while (s < le && isspace((unsigned char)*s)) ++s;
            while (le > s && isspace((unsigned char)le[-1])) --le;
            if (le > s) { out = {s, le, no}; return true; }
        }
        return false;
    }
};

int split_ws(const Line& l, const char* tb[8], const char* te[8]) {
    int n = 0;
    const char* s = l.b;
    while (s < l.e) {
        while (s < l.e && isspace((unsigned char)*s)) ++s;
        if (s >= l.e) break;
        const char* t = s;
        while (t < l.e && !isspace((unsigned char)*t)) ++t;
        if (n < 8) { tb[n] = s; te[n] = t; }
        ++n;
        s = t;
    }
    return n;
}

bool parse_usize(const char* b, const char* e, uint64_t& out) {  // Rust: optional '+', digits only
    if (b < e && *b == '+') ++b;
    if (b >= e) return false;
    uint64_t v = 0;
    for (const char* c = b; c < e; ++c) {
        if (*c < '0' || *c > '9') return false;
        if (v > (UINT64_MAX - 9) / 10) return false;
        v = v * 10 + (uint64_t)(*c - '0');
    }
    out = v;
    return true;
}

bool parse_f64(const char* b, const char* e, double& out) {
    char buf[128];
    const size_t n = (size_t)(e - b);
    if (n == 0 || n >= sizeof buf) return false;
    memcpy(buf, b, n); buf[n] = 0;
    if (buf[0] == '0' && (buf[1] == 'x' || buf[1] == 'X')) return false;  // Rust has no hex floats
    char* endp = nullptr;
    errno = 0;
    out = strtod(buf, &endp);
    return endp == buf + n;
}
}  // namespace

extern "C" {

const char* apexgpu_bal_last_error(void) { return g_bal_err.c_str(); }

static int bal_open_impl(const char* path, apexgpu_bal** out);
// No exception crosses the C boundary: allocation failures (sizes come from an untrusted header) are a parse error.
int apexgpu_bal_open(const char* path, apexgpu_bal** out) {
    try {
        return bal_open_impl(path, out);
    } catch (const std::exception& e) {
        if (out) *out = nullptr;
        return bal_fail(APEXGPU_BAL_ERR_PARSE, std::string("BAL file too large for this host: ") + e.what());
    } catch (...) {
        if (out) *out = nullptr;
        return bal_fail(APEXGPU_BAL_ERR_PARSE, "unexpected failure while reading the BAL file");
    }
}
}  // extern "C"

static int bal_open_impl(const char* path, apexgpu_bal** out) {
    if (!path || !out) return bal_fail(APEXGPU_ERR_INVALID_INPUT, "null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return bal_fail(APEXGPU_BAL_ERR_IO, std::string("Failed to read BAL file: ") + path + ": " + strerror(errno));
    std::string content;
    {
        fseek(f, 0, SEEK_END);
        long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        if (sz > 0) { content.resize((size_t)sz); if (fread(&content[0], 1, (size_t)sz, f) != (size_t)sz) { fclose(f); return bal_fail(APEXGPU_BAL_ERR_IO, "short read"); } }
        fclose(f);
    }
    LineIter it{content.data(), content.data() + content.size()};
    Line l;
    const char *tb[8], *te[8];
    std::unique_ptr<apexgpu_bal> ds(new apexgpu_bal());
    auto bad = [&](int code, const std::string& m) { return bal_fail(code, m); };
    // header (bal.rs:205-241)
    if (!it.next(l)) return bad(APEXGPU_BAL_ERR_PARSE, "line 1: Missing header line");
    if (split_ws(l, tb, te) != 3) return bad(APEXGPU_BAL_ERR_MISSING_FIELDS, "line " + std::to_string(l.no) + ": missing fields");
    uint64_t hc, hp, ho;
    if (!parse_usize(tb[0], te[0], hc)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[0], te[0]) + "'");
    if (!parse_usize(tb[1], te[1], hp)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[1], te[1]) + "'");
    if (!parse_usize(tb[2], te[2], ho)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[2], te[2]) + "'");
    // The header is untrusted: every observation line needs >= 8 bytes ("0 0 0 0\n"), every camera / point value >= 2,
    // so counts the file cannot hold are the reference's "Unexpected end of file" before anything is reserved.
    // (The reference's Vec::with_capacity(num_observations) panics on such a header, bal.rs:247.)
    const uint64_t fsz = (uint64_t)content.size();
    if (ho > fsz / 8 + 1) return bad(APEXGPU_BAL_ERR_PARSE, "Unexpected end of file in observations section");
    if (hc > fsz / 18 + 1) return bad(APEXGPU_BAL_ERR_PARSE, "Unexpected end of file in camera 0 parameter 0");
    if (hp > fsz / 6 + 1) return bad(APEXGPU_BAL_ERR_PARSE, "Unexpected end of file in point 0 coordinate 0");
    ds->n_cam = (int64_t)hc; ds->n_pt = (int64_t)hp; ds->n_obs = (int64_t)ho;
    // observations (:243-300)
    ds->cam_idx.reserve(ho); ds->pt_idx.reserve(ho); ds->obs_uv.reserve(2 * ho);
    for (uint64_t i = 0; i < ho; ++i) {
        if (!it.next(l)) return bad(APEXGPU_BAL_ERR_PARSE, "Unexpected end of file in observations section");
        if (split_ws(l, tb, te) != 4) return bad(APEXGPU_BAL_ERR_MISSING_FIELDS, "line " + std::to_string(l.no) + ": missing fields");
        uint64_t c, p; double x, y;
        if (!parse_usize(tb[0], te[0], c)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[0], te[0]) + "'");
        if (!parse_usize(tb[1], te[1], p)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[1], te[1]) + "'");
        if (!parse_f64(tb[2], te[2], x)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[2], te[2]) + "'");
        if (!parse_f64(tb[3], te[3], y)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[3], te[3]) + "'");
        // indices are stored as u32 (the C ABI's index type); the reference keeps usize and never range-checks them in
        // the loader, so an index >= the header count still loads (set_structure rejects it), but one that does not fit
        // u32 would alias a valid index here: refuse it as an invalid number
        if (c > 0xFFFFFFFFull) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[0], te[0]) + "' (camera index exceeds 32 bits)");
        if (p > 0xFFFFFFFFull) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(tb[1], te[1]) + "' (point index exceeds 32 bits)");
        ds->cam_idx.push_back((uint32_t)c); ds->pt_idx.push_back((uint32_t)p);
        ds->obs_uv.push_back(x); ds->obs_uv.push_back(y);
    }
    // cameras: 9 values, one per line (:302-345)
    ds->cam_raw.reserve(9 * hc);
    for (uint64_t c = 0; c < hc; ++c)
        for (int k = 0; k < 9; ++k) {
            if (!it.next(l)) return bad(APEXGPU_BAL_ERR_PARSE, "Unexpected end of file in camera " + std::to_string(c) + " parameter " + std::to_string(k));
            double v;
            if (!parse_f64(l.b, l.e, v)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(l.b, l.e) + "'");
            if (k == 6 && !(v > 0.0 && std::isfinite(v))) v = 500.0;  // normalize_focal_length (:100-114)
            ds->cam_raw.push_back(v);
        }
    // points: 3 values, one per line (:347-388)
    ds->points.reserve(3 * hp);
    for (uint64_t p = 0; p < hp; ++p)
        for (int k = 0; k < 3; ++k) {
            if (!it.next(l)) return bad(APEXGPU_BAL_ERR_PARSE, "Unexpected end of file in point " + std::to_string(p) + " coordinate " + std::to_string(k));
            double v;
            if (!parse_f64(l.b, l.e, v)) return bad(APEXGPU_BAL_ERR_INVALID_NUMBER, "line " + std::to_string(l.no) + ": invalid number '" + std::string(l.b, l.e) + "'");
            ds->points.push_back(v);
        }
    *out = ds.release();
    return APEXGPU_OK;
}

extern "C" {

void apexgpu_bal_close(apexgpu_bal* b) { delete b; }

int apexgpu_bal_sizes(const apexgpu_bal* b, int64_t* n_cam, int64_t* n_pt, int64_t* n_obs) {
    if (!b) return APEXGPU_ERR_INVALID_INPUT;
    if (n_cam) *n_cam = b->n_cam;
    if (n_pt) *n_pt = b->n_pt;
    if (n_obs) *n_obs = b->n_obs;
    return APEXGPU_OK;
}

int apexgpu_bal_raw(const apexgpu_bal* b, uint32_t* cam_idx, uint32_t* pt_idx, double* obs_uv, double* cameras9,
                    double* points3) {
    if (!b) return APEXGPU_ERR_INVALID_INPUT;
    if (cam_idx) memcpy(cam_idx, b->cam_idx.data(), b->cam_idx.size() * 4);
    if (pt_idx) memcpy(pt_idx, b->pt_idx.data(), b->pt_idx.size() * 4);
    if (obs_uv) memcpy(obs_uv, b->obs_uv.data(), b->obs_uv.size() * 8);
    if (cameras9) memcpy(cameras9, b->cam_raw.data(), b->cam_raw.size() * 8);
    if (points3) memcpy(points3, b->points.data(), b->points.size() * 8);
    return APEXGPU_OK;
}

// The variables run_bundle_adjustment builds (bin/bundle_adjustment.rs:200-208, 232-257):
// pose_i = SE3(translation, axis-angle -> unit quaternion) as [tx,ty,tz,qw,qx,qy,qz], intr_i = [f,k1,k2].
int apexgpu_bal_variables(const apexgpu_bal* b, double* poses7, double* intr3) {
    if (!b || !poses7 || !intr3) return APEXGPU_ERR_INVALID_INPUT;
    for (int64_t c = 0; c < b->n_cam; ++c) {
        const double* r = b->cam_raw.data() + 9 * c;
        const double angle = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
        double q[4] = {1.0, 0.0, 0.0, 0.0};
        if (!(angle < 1e-10)) {  // axis_angle_to_so3: identity below 1e-10, else from_axis_angle(axis, angle)
            double ax[3] = {r[0] / angle, r[1] / angle, r[2] / angle};
            const double an = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);  // Unit::new_normalize
            const double s = sin(angle / 2.0);
            q[0] = cos(angle / 2.0); q[1] = ax[0] / an * s; q[2] = ax[1] / an * s; q[3] = ax[2] / an * s;
        }
        double* p = poses7 + 7 * c;
        p[0] = r[3]; p[1] = r[4]; p[2] = r[5]; p[3] = q[0]; p[4] = q[1]; p[5] = q[2]; p[6] = q[3];
        intr3[3 * c] = r[6]; intr3[3 * c + 1] = r[7]; intr3[3 * c + 2] = r[8];
    }
    return APEXGPU_OK;
}

// First global column of intr_{i:04} / pose_{i:04} / pt_{j:05} when all names are sorted as byte
// strings (src/optimizer/mod.rs:530-536): [intr_* | pose_* | pt_*], zero-padded decimal order inside.
int apexgpu_reference_columns(int64_t n_cam, int64_t n_pt, int64_t* intr_col, int64_t* pose_col, int64_t* pt_col) {
    if (n_cam < 0 || n_pt < 0 || !intr_col || !pose_col || !pt_col) return APEXGPU_ERR_INVALID_INPUT;
    auto lex_rank = [](int64_t n, int width, std::vector<int64_t>& rank) {
        rank.resize(n);
        int64_t lim = 1;
        for (int i = 0; i < width; ++i) lim *= 10;
        if (n <= lim) { for (int64_t i = 0; i < n; ++i) rank[i] = i; return; }
        std::vector<std::string> names(n);
        char buf[32];
        for (int64_t i = 0; i < n; ++i) { snprintf(buf, sizeof buf, "%0*lld", width, (long long)i); names[i] = buf; }
        std::vector<int64_t> order(n);
        for (int64_t i = 0; i < n; ++i) order[i] = i;
        std::sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return names[a] < names[b]; });
        for (int64_t pos = 0; pos < n; ++pos) rank[order[pos]] = pos;
    };
    std::vector<int64_t> cr, pr;
    lex_rank(n_cam, 4, cr);
    lex_rank(n_pt, 5, pr);
    for (int64_t c = 0; c < n_cam; ++c) { intr_col[c] = 3 * cr[c]; pose_col[c] = 3 * n_cam + 6 * cr[c]; }
    for (int64_t j = 0; j < n_pt; ++j) pt_col[j] = 9 * n_cam + 3 * pr[j];
    return APEXGPU_OK;
}

}  // extern "C"
