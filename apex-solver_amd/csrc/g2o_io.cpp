// g2o_io.cpp -- host-side input path of the pose-graph backend: the G2O text reader and the
// problem builder of bin/pose_graph_g2o.rs, in C++.  No device code.
//
// Reference (file:line under the apex-solver tree):
//   G2oLoader::load / parse_content / parse_line          crates/apex-io/src/g2o.rs:140-300
//   parse_vertex_se3 (norm check |n-1| <= 0.01, normalise) :340-420
//   parse_edge_se3 (21 upper-triangular information values) :488-620
//   parse_vertex_se2 / parse_edge_se2 (validated, SE2 payload not kept: out of scope)  :303-337, 424-485
//   variables "x{id}" sorted by id, column order = sorted names, first vertex fixed
//                                                          bin/pose_graph_g2o.rs:748-797, 828-838
#include "../../include/apexgpu.h"

#include <ctype.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_set>
#include <vector>

struct apexgpu_g2o {
    std::vector<int64_t> ids;       // SE3 vertex ids in file order
    std::vector<double> poses;      // 7 per vertex [t, qw,qx,qy,qz]
    std::vector<int64_t> e_from, e_to;  // vertex ids
    std::vector<double> meas;       // 7 per edge
    std::vector<double> info;       // 36 per edge (row-major symmetric)
    int64_t n_se2_vertices = 0, n_se2_edges = 0;
};

static thread_local std::string g_g2o_err;
static int g2o_fail(int code, const std::string& msg) { g_g2o_err = msg; return code; }

namespace {

struct Tok { const char* b; const char* e; };

bool tok_usize(const Tok& t, uint64_t& out) {  // Rust usize::from_str: optional '+', digits only
    const char* b = t.b;
    if (b < t.e && *b == '+') ++b;
    if (b >= t.e) return false;
    uint64_t v = 0;
    for (const char* c = b; c < t.e; ++c) {
        if (*c < '0' || *c > '9') return false;
        if (v > (UINT64_MAX - 9) / 10) return false;
        v = v * 10 + (uint64_t)(*c - '0');
    }
    out = v;
    return true;
}

bool tok_f64(const Tok& t, double& out) {  // Rust f64::from_str: decimal / exponent, inf, infinity, nan; no hex, no spaces
    const size_t n = (size_t)(t.e - t.b);
    if (n == 0 || n > 400) return false;
    char buf[408];
    memcpy(buf, t.b, n);
    buf[n] = 0;
    const char* p = buf;
    if (*p == '+' || *p == '-') ++p;
    if (*p == 0) return false;
    auto ieq = [](const char* a, const char* b) {
        for (; *a && *b; ++a, ++b)
            if (tolower((unsigned char)*a) != *b) return false;
        return *a == 0 && *b == 0;
    };
    if (ieq(p, "inf") || ieq(p, "infinity") || ieq(p, "nan")) {
        out = strtod(buf, nullptr);
        return true;
    }
    bool digits = false;
    for (const char* c = p; *c; ++c) {
        if (*c >= '0' && *c <= '9') { digits = true; continue; }
        if (*c == '.' || *c == 'e' || *c == 'E' || *c == '+' || *c == '-') continue;
        return false;
    }
    if (!digits) return false;
    char* endp = nullptr;
    out = strtod(buf, &endp);
    return endp && *endp == 0;
}

std::string tok_str(const Tok& t) { return std::string(t.b, t.e); }

int invalid_number(size_t line, const Tok& t) {
    return g2o_fail(APEXGPU_G2O_ERR_INVALID_NUMBER, "Invalid number format at line " + std::to_string(line) + ": " + tok_str(t));
}
int missing_fields(size_t line) {
    return g2o_fail(APEXGPU_G2O_ERR_MISSING_FIELDS, "Missing required fields at line " + std::to_string(line));
}

}  // namespace

extern "C" {

const char* apexgpu_g2o_last_error(void) { return g_g2o_err.c_str(); }

void apexgpu_g2o_close(apexgpu_g2o* g) { delete g; }

static int g2o_open_impl(const char* path, apexgpu_g2o** out);
int apexgpu_g2o_open(const char* path, apexgpu_g2o** out) {  // no exception crosses the C boundary
    try {
        return g2o_open_impl(path, out);
    } catch (const std::exception& e) {
        if (out) *out = nullptr;
        return g2o_fail(APEXGPU_G2O_ERR_IO, std::string("IO error: out of memory while reading (") + e.what() + ")");
    } catch (...) {
        if (out) *out = nullptr;
        return g2o_fail(APEXGPU_G2O_ERR_IO, "IO error: unexpected failure while reading");
    }
}
}  // extern "C"
static int g2o_open_impl(const char* path, apexgpu_g2o** out) {
    if (!path || !out) return g2o_fail(APEXGPU_G2O_ERR_IO, "IO error: null argument");
    *out = nullptr;
    FILE* f = fopen(path, "rb");
    if (!f) return g2o_fail(APEXGPU_G2O_ERR_IO, std::string("IO error: ") + strerror(errno) + " (" + path + ")");
    std::string buf;
    {
        char chunk[1 << 16];
        size_t n;
        while ((n = fread(chunk, 1, sizeof chunk, f)) > 0) buf.append(chunk, n);
        fclose(f);
    }
    std::unique_ptr<apexgpu_g2o> gown(new apexgpu_g2o());
    apexgpu_g2o* g = gown.get();
    std::unordered_set<int64_t> seen3, seen2;
    std::vector<Tok> parts;
    const char* p = buf.data();
    const char* end = p + buf.size();
    size_t line_no = 0;
    int rc = 0;
    while (p < end && rc == 0) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(end - p));
        const char* le = nl ? nl : end;
        const char* s = p;
        p = nl ? nl + 1 : end;
        ++line_no;
        while (s < le && isspace((unsigned char)*s)) ++s;
        while (le > s && isspace((unsigned char)le[-1])) --le;
        if (s >= le || *s == '#') continue;  // empty lines and comments (g2o.rs:235-238)
        parts.clear();
        for (const char* c = s; c < le;) {
            while (c < le && isspace((unsigned char)*c)) ++c;
            if (c >= le) break;
            const char* t = c;
            while (t < le && !isspace((unsigned char)*t)) ++t;
            parts.push_back({c, t});
            c = t;
        }
        if (parts.empty()) continue;
        const std::string tag = tok_str(parts[0]);
        if (tag == "VERTEX_SE3:QUAT") {
            if (parts.size() < 9) { rc = missing_fields(line_no); break; }
            uint64_t id;
            if (!tok_usize(parts[1], id)) { rc = invalid_number(line_no, parts[1]); break; }
            double v[7];  // x y z qx qy qz qw
            for (int k = 0; k < 7 && rc == 0; ++k)
                if (!tok_f64(parts[2 + k], v[k])) rc = invalid_number(line_no, parts[2 + k]);
            if (rc) break;
            const double qx = v[3], qy = v[4], qz = v[5], qw = v[6];
            const double norm = sqrt(qw * qw + qx * qx + qy * qy + qz * qz);
            if (fabs(norm - 1.0) > 0.01) {
                char msg[160];
                snprintf(msg, sizeof msg, "Invalid quaternion at line %zu: norm = %.6f, expected ~1.0", line_no, norm);
                rc = g2o_fail(APEXGPU_G2O_ERR_INVALID_QUATERNION, msg);
                break;
            }
            if (!seen3.insert((int64_t)id).second) {
                rc = g2o_fail(APEXGPU_G2O_ERR_DUPLICATE_VERTEX, "Duplicate vertex ID: " + std::to_string(id));
                break;
            }
            // "always normalize for numerical safety", then from_translation_quaternion normalises again
            double w = qw, x = qx, y = qy, z = qz;
            for (int pass = 0; pass < 2; ++pass) {
                const double n = sqrt(w * w + x * x + y * y + z * z);
                w /= n; x /= n; y /= n; z /= n;
            }
            g->ids.push_back((int64_t)id);
            const double pose[7] = {v[0], v[1], v[2], w, x, y, z};
            g->poses.insert(g->poses.end(), pose, pose + 7);
        } else if (tag == "EDGE_SE3:QUAT") {
            if (parts.size() < 10) { rc = missing_fields(line_no); break; }
            uint64_t from, to;
            if (!tok_usize(parts[1], from)) { rc = invalid_number(line_no, parts[1]); break; }
            if (!tok_usize(parts[2], to)) { rc = invalid_number(line_no, parts[2]); break; }
            double v[7];  // tx ty tz qx qy qz qw
            for (int k = 0; k < 7 && rc == 0; ++k)
                if (!tok_f64(parts[3 + k], v[k])) rc = invalid_number(line_no, parts[3 + k]);
            if (rc) break;
            // the reference indexes parts[10..31] unconditionally; a shorter line is reported, not a panic
            if (parts.size() < 31) {
                rc = g2o_fail(APEXGPU_G2O_ERR_PARSE, "Parse error at line " + std::to_string(line_no) + ": Invalid information matrix values");
                break;
            }
            double iv[21];
            for (int k = 0; k < 21 && rc == 0; ++k)
                if (!tok_f64(parts[10 + k], iv[k]))
                    rc = g2o_fail(APEXGPU_G2O_ERR_PARSE, "Parse error at line " + std::to_string(line_no) + ": Invalid information matrix values");
            if (rc) break;
            double w = v[6], x = v[3], y = v[4], z = v[5];
            {  // UnitQuaternion::from_quaternion
                const double n = sqrt(w * w + x * x + y * y + z * z);
                w /= n; x /= n; y /= n; z /= n;
            }
            g->e_from.push_back((int64_t)from);
            g->e_to.push_back((int64_t)to);
            const double m[7] = {v[0], v[1], v[2], w, x, y, z};
            g->meas.insert(g->meas.end(), m, m + 7);
            double I[36];
            int k = 0;
            for (int i = 0; i < 6; ++i)
                for (int j = i; j < 6; ++j) { I[6 * i + j] = iv[k]; I[6 * j + i] = iv[k]; ++k; }
            g->info.insert(g->info.end(), I, I + 36);
        } else if (tag == "VERTEX_SE2") {
            if (parts.size() < 5) { rc = missing_fields(line_no); break; }
            uint64_t id;
            if (!tok_usize(parts[1], id)) { rc = invalid_number(line_no, parts[1]); break; }
            double d;
            for (int k = 2; k < 5 && rc == 0; ++k)
                if (!tok_f64(parts[k], d)) rc = invalid_number(line_no, parts[k]);
            if (rc) break;
            if (!seen2.insert((int64_t)id).second) {
                rc = g2o_fail(APEXGPU_G2O_ERR_DUPLICATE_VERTEX, "Duplicate vertex ID: " + std::to_string(id));
                break;
            }
            g->n_se2_vertices++;
        } else if (tag == "EDGE_SE2") {
            if (parts.size() < 12) { rc = missing_fields(line_no); break; }
            uint64_t id;
            if (!tok_usize(parts[1], id)) { rc = invalid_number(line_no, parts[1]); break; }
            if (!tok_usize(parts[2], id)) { rc = invalid_number(line_no, parts[2]); break; }
            double d;
            for (int k = 3; k < 6 && rc == 0; ++k)
                if (!tok_f64(parts[k], d)) rc = invalid_number(line_no, parts[k]);
            for (int k = 6; k < 12 && rc == 0; ++k)
                if (!tok_f64(parts[k], d))
                    rc = g2o_fail(APEXGPU_G2O_ERR_PARSE, "Parse error at line " + std::to_string(line_no) + ": Invalid information matrix values");
            if (rc) break;
            g->n_se2_edges++;
        }
        // unknown tags are skipped silently (g2o.rs:268-270)
    }
    if (rc != 0) return rc;
    *out = gown.release();
    return 0;
}

extern "C" {

int apexgpu_g2o_sizes(const apexgpu_g2o* g, int64_t* n_vertices_se3, int64_t* n_edges_se3, int64_t* n_vertices_se2,
                      int64_t* n_edges_se2) {
    if (!g) return g2o_fail(APEXGPU_G2O_ERR_IO, "null handle");
    if (n_vertices_se3) *n_vertices_se3 = (int64_t)g->ids.size();
    if (n_edges_se3) *n_edges_se3 = (int64_t)g->e_from.size();
    if (n_vertices_se2) *n_vertices_se2 = g->n_se2_vertices;
    if (n_edges_se2) *n_edges_se2 = g->n_se2_edges;
    return 0;
}

int apexgpu_g2o_raw(const apexgpu_g2o* g, int64_t* ids, double* poses7, int64_t* e_from, int64_t* e_to, double* meas7,
                    double* info36) {
    if (!g) return g2o_fail(APEXGPU_G2O_ERR_IO, "null handle");
    if (ids) memcpy(ids, g->ids.data(), g->ids.size() * sizeof(int64_t));
    if (poses7) memcpy(poses7, g->poses.data(), g->poses.size() * sizeof(double));
    if (e_from) memcpy(e_from, g->e_from.data(), g->e_from.size() * sizeof(int64_t));
    if (e_to) memcpy(e_to, g->e_to.data(), g->e_to.size() * sizeof(int64_t));
    if (meas7) memcpy(meas7, g->meas.data(), g->meas.size() * sizeof(double));
    if (info36) memcpy(info36, g->info.data(), g->info.size() * sizeof(double));
    return 0;
}

/* The problem bin/pose_graph_g2o.rs builds: vertices sorted by id (variable "x{id}"), edges in file
 * order with endpoints as indices into that sorted list, the first vertex fixed (LM gauge). */
int apexgpu_g2o_problem(const apexgpu_g2o* g, int64_t* sorted_ids, double* poses7, uint32_t* e_from, uint32_t* e_to,
                        double* meas7, int64_t* pose_col, uint8_t* fix6) {
    if (!g) return g2o_fail(APEXGPU_G2O_ERR_IO, "null handle");
    const size_t nv = g->ids.size(), ne = g->e_from.size();
    std::vector<size_t> order(nv);
    for (size_t i = 0; i < nv; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return g->ids[a] < g->ids[b]; });
    std::vector<int64_t> ids(nv);
    for (size_t k = 0; k < nv; ++k) {
        ids[k] = g->ids[order[k]];
        if (sorted_ids) sorted_ids[k] = ids[k];
        if (poses7) memcpy(poses7 + 7 * k, g->poses.data() + 7 * order[k], 7 * sizeof(double));
    }
    for (size_t e = 0; e < ne; ++e) {
        const auto fa = std::lower_bound(ids.begin(), ids.end(), g->e_from[e]);
        const auto fb = std::lower_bound(ids.begin(), ids.end(), g->e_to[e]);
        if (fa == ids.end() || *fa != g->e_from[e] || fb == ids.end() || *fb != g->e_to[e])
            return g2o_fail(APEXGPU_G2O_ERR_PARSE, "edge " + std::to_string(e) + " references a vertex that is not in the file");
        if (e_from) e_from[e] = (uint32_t)(fa - ids.begin());
        if (e_to) e_to[e] = (uint32_t)(fb - ids.begin());
    }
    if (meas7) memcpy(meas7, g->meas.data(), g->meas.size() * sizeof(double));
    if (pose_col) {
        const int rc = apexgpu_pose_graph_columns((int64_t)nv, ids.data(), pose_col);
        if (rc != 0) return rc;
    }
    if (fix6) {
        memset(fix6, 0, 6 * nv);
        if (nv > 0) memset(fix6, 1, 6);  // fix_variable(x{first}, 0..5) (pose_graph_g2o.rs:790-797)
    }
    return 0;
}

/* First global column of "x{id}" in the sorted-name order of src/optimizer/mod.rs:530-536
 * (names compare as strings: x0, x1, x10, x100, ..., x2, ...). */
int apexgpu_pose_graph_columns(int64_t n_v, const int64_t* ids, int64_t* pose_col) {
    if (n_v < 0 || (n_v > 0 && (!ids || !pose_col))) return g2o_fail(APEXGPU_G2O_ERR_PARSE, "bad arguments");
    std::vector<std::pair<std::string, int64_t>> names((size_t)n_v);
    for (int64_t v = 0; v < n_v; ++v) names[(size_t)v] = {"x" + std::to_string(ids[v]), v};
    std::sort(names.begin(), names.end());
    for (int64_t r = 0; r < n_v; ++r) pose_col[names[(size_t)r].second] = 6 * r;
    return 0;
}

}  // extern "C"
