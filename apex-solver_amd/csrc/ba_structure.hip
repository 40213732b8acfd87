// ba_structure.hip -- see ba_structure.h.  Host code only.
#include "ba_structure.h"

#include <math.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <mutex>
#include <numeric>

#include "host_parallel.h"

namespace apex {

void shard_range(int64_t n_pt, const int64_t* ptr, int rank, int world, int64_t* lo, int64_t* hi);  // solver.hip

namespace {
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
}  // namespace

std::string BaHostStructure::build_lists(int64_t n_cam_, int64_t n_pt_, int64_t n_obs_, const uint32_t* cam_idx,
                                         const uint32_t* pt_idx, const double* obs_uv, const BaStructOptions& o, TilePlan& tp) {
    build_order(n_cam_, n_pt_, n_obs_, cam_idx, pt_idx, o, tp);
    return build_obs_lists(cam_idx, pt_idx, obs_uv, o, tp);
}

void BaHostStructure::build_order(int64_t n_cam_, int64_t n_pt_, int64_t n_obs_, const uint32_t* cam_idx,
                                  const uint32_t* pt_idx, const BaStructOptions& o, TilePlan& tp) {
    const double t_begin = now_s();
    SetupTrace tr;
    n_cam = n_cam_; n_pt = n_pt_; n_obs = n_obs_; dc = o.dc;
    n_c = n_cam * dc;
    nt = (int)((n_c + kNB - 1) / kNB);
    n_c_pad = (int64_t)nt * kNB;
    const int cpt = kNB / dc;
    const int rank = o.rank, world = o.world;

    // ---- landmark-major view of the full problem in the caller's numbering (counting sort) ------------------------------
    std::vector<int64_t>& lp = lp_;
    raw_vector<int>& lobs = lobs_;
    parallel_bucket_large(n_obs, n_pt, pt_idx, lp, lobs);
    tr.mark("order: landmark buckets");

    // weight of a tile pair = number of (landmark, tile pair) incidences: thousands for tiles that overlap in a capture
    // sequence, one or two for an accidental long-range match.  An edge of the tile graph is STRONG when it carries at
    // least strong_edge_frac of the heaviest off-diagonal weight of its weaker end.
    auto tile_weights = [&](const std::vector<int>& pos, std::vector<uint32_t>& acnt, std::vector<uint8_t>& strong) {
        acnt.assign((size_t)nt * nt, 0);
        if ((size_t)nt * nt <= ((size_t)2 << 20)) {
            // few tiles (final-13682: 856^2 counters = 2.9 MB): every thread counts into a table of its own -- plain increments
            // that stay in its cache -- and the tables are added at the end; the sort + run-length form below was 45 of the
            // ordering's 75 ms (round 5)
            const int nth = std::max(1, std::min<int>(host_threads(), (int)(n_pt / 65536) + 1));
            std::vector<std::vector<uint32_t>> local((size_t)nth);
            const int64_t per = (n_pt + nth - 1) / nth;
            parallel_rows(nth, [&](int64_t t) {
                std::vector<uint32_t>& tab = local[(size_t)t];
                tab.assign((size_t)nt * nt, 0);
                int tl[64];
                std::vector<int> big;
                for (int64_t l = t * per; l < std::min<int64_t>(n_pt, (t + 1) * per); ++l) {
                    const int64_t k = lp[l + 1] - lp[l];
                    int* q = tl;
                    if (k > 64) { big.resize((size_t)k); q = big.data(); }
                    int n = 0;
                    for (int64_t x = lp[l]; x < lp[l + 1]; ++x) {   // distinct tiles of the landmark (a handful: insertion into a sorted run)
                        const int tv = pos[cam_idx[lobs[x]]] / cpt;
                        int a = 0;
                        while (a < n && q[a] < tv) ++a;
                        if (a < n && q[a] == tv) continue;
                        for (int m = n; m > a; --m) q[m] = q[m - 1];
                        q[a] = tv; ++n;
                    }
                    for (int a = 0; a < n; ++a)
                        for (int bb = 0; bb <= a; ++bb) tab[(size_t)q[a] * nt + q[bb]]++;
                }
            }, 1);
            parallel_ranges((int64_t)nt * nt, 1 << 16, [&](int64_t b, int64_t e) {
                for (int t = 0; t < nth; ++t) {
                    const uint32_t* tab = local[(size_t)t].data();
                    for (int64_t i = b; i < e; ++i) acnt[(size_t)i] += tab[i];
                }
            });
        } else
        parallel_ranges(n_pt, 4096, [&](int64_t b, int64_t e) {
            // the incidences of a landmark range fall on a handful of tile pairs: count them locally (sort + run
            // lengths) and add every distinct pair ONCE -- per-incidence atomics from all threads on the same few
            // counters were two thirds of the ordering time
            std::vector<int> tl;
            std::vector<uint32_t> keys;
            keys.reserve(32768);
            for (int64_t l = b; l < e; ++l) {
                tl.clear();
                for (int64_t x = lp[l]; x < lp[l + 1]; ++x) tl.push_back(pos[cam_idx[lobs[x]]] / cpt);
                std::sort(tl.begin(), tl.end());
                tl.erase(std::unique(tl.begin(), tl.end()), tl.end());
                for (size_t a = 0; a < tl.size(); ++a)
                    for (size_t bb = 0; bb <= a; ++bb) keys.push_back((uint32_t)tl[a] * (uint32_t)nt + (uint32_t)tl[bb]);
            }
            std::sort(keys.begin(), keys.end());
            for (size_t i = 0; i < keys.size();) {
                size_t j = i;
                while (j < keys.size() && keys[j] == keys[i]) ++j;
                __atomic_fetch_add(&acnt[keys[i]], (uint32_t)(j - i), __ATOMIC_RELAXED);
                i = j;
            }
        });
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < a; ++b) acnt[(size_t)b * nt + a] = acnt[(size_t)a * nt + b];
        std::vector<uint32_t> rowmax(nt, 0);
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < nt; ++b)
                if (a != b) rowmax[a] = std::max(rowmax[a], acnt[(size_t)a * nt + b]);
        strong.assign((size_t)nt * nt, 0);
        for (int a = 0; a < nt; ++a)
            for (int b = 0; b < nt; ++b) {
                const uint32_t w = acnt[(size_t)a * nt + b];
                if (w == 0) continue;
                const double need = o.strong_edge_frac * (double)std::min(rowmax[a], rowmax[b]);
                if (a == b || (double)w >= need) strong[(size_t)a * nt + b] = 1;
            }
    };

    // ---- border cameras ------------------------------------------------------------------------------------------------------
    // A tile Cholesky wants a tile graph with geometry: bands, trees, grids.  Two things in real captures destroy it at
    // tile granularity although they involve few CAMERAS: hub cameras (an overview photograph covisible with a large
    // part of the collection makes its whole 16-camera tile a dense row of S) and accidental long-range matches (one
    // camera far away sees a landmark of this neighbourhood: its tile couples with the four or five tiles of the
    // landmark's window).  Both show up as camera pairs that share a landmark while their tiles are NOT strongly
    // connected.  A greedy vertex cover of those pairs (highest degree first: hubs go first) is ordered LAST: S becomes
    // the geometric part plus a dense border.  The order is a heuristic; the fill below is computed on the true structure.
    std::vector<int> pre(n_cam);
    std::iota(pre.begin(), pre.end(), 0);
    n_hubs = 0;
    std::vector<uint32_t> acnt;
    std::vector<uint8_t> adjm;
    tile_weights(pre, acnt, adjm);
    tr.mark("order: tile weights");
    if (o.hubs_last && nt >= 24) {
        // "not strongly connected" = more than two strong hops apart: the far ends of a capture window (tiles that
        // overlap in a few landmarks only but have common strong neighbours) are geometry, not long-range matches
        std::vector<uint8_t> near2((size_t)nt * nt, 0);
        {
            const int words = (nt + 63) / 64;
            std::vector<uint64_t> rowbits((size_t)nt * words, 0);
            for (int a = 0; a < nt; ++a)
                for (int b = 0; b < nt; ++b)
                    if (adjm[(size_t)a * nt + b]) rowbits[(size_t)a * words + (b >> 6)] |= 1ull << (b & 63);
            parallel_rows(nt, [&](int64_t a) {
                std::vector<uint64_t> acc(rowbits.begin() + a * words, rowbits.begin() + (a + 1) * words);
                for (int b = 0; b < nt; ++b)
                    if (adjm[(size_t)a * nt + b])
                        for (int w = 0; w < words; ++w) acc[w] |= rowbits[(size_t)b * words + w];
                for (int b = 0; b < nt; ++b) near2[(size_t)a * nt + b] = (acc[b >> 6] >> (b & 63)) & 1;
            });
        }
        tr.mark("order: two-hop closure");
        std::vector<uint64_t> edges;
        std::mutex edges_mu;
        std::atomic<int64_t> total(0);
        const int64_t kMaxEdges = 1LL << 26;
        parallel_ranges(n_pt, 4096, [&](int64_t b, int64_t e) {
            if (total.load(std::memory_order_relaxed) > kMaxEdges) return;
            std::vector<uint64_t> out;
            for (int64_t l = b; l < e; ++l)
                for (int64_t x = lp[l]; x < lp[l + 1]; ++x) {
                    const uint32_t ca = cam_idx[lobs[x]];
                    for (int64_t y = x + 1; y < lp[l + 1]; ++y) {
                        const uint32_t cb = cam_idx[lobs[y]];
                        if (ca == cb || near2[(size_t)(ca / cpt) * nt + cb / cpt]) continue;
                        out.push_back(((uint64_t)std::min(ca, cb) << 32) | std::max(ca, cb));
                    }
                }
            if (out.empty()) return;
            total.fetch_add((int64_t)out.size(), std::memory_order_relaxed);
            std::lock_guard<std::mutex> lk(edges_mu);
            if ((int64_t)edges.size() <= kMaxEdges) edges.insert(edges.end(), out.begin(), out.end());
        });
        tr.mark("order: long-range pairs");
        if (total.load() <= kMaxEdges && total.load() > 0) {
            std::sort(edges.begin(), edges.end());
            edges.erase(std::unique(edges.begin(), edges.end()), edges.end());
            // adjacency of the long-range graph
            std::vector<int> deg(n_cam, 0);
            for (uint64_t k : edges) { deg[k >> 32]++; deg[(uint32_t)k]++; }
            std::vector<int64_t> ap(n_cam + 1, 0);
            for (int64_t c = 0; c < n_cam; ++c) ap[c + 1] = ap[c] + deg[c];
            std::vector<int> an(ap[n_cam]);
            {
                std::vector<int64_t> fill(ap.begin(), ap.end() - 1);
                for (uint64_t k : edges) { const int a = (int)(k >> 32), b = (int)(uint32_t)k; an[fill[a]++] = b; an[fill[b]++] = a; }
            }
            // greedy cover, highest remaining degree first (lazy max-heap)
            std::vector<std::pair<int, int>> heap;
            for (int64_t c = 0; c < n_cam; ++c) if (deg[c] > 0) heap.push_back({deg[c], (int)c});
            std::make_heap(heap.begin(), heap.end());
            std::vector<char> in_cover(n_cam, 0);
            std::vector<int> cover;
            int64_t alive = (int64_t)edges.size();
            while (alive > 0 && !heap.empty()) {
                std::pop_heap(heap.begin(), heap.end());
                const std::pair<int, int> top = heap.back();
                heap.pop_back();
                const int c = top.second;
                if (in_cover[c] || top.first != deg[c] || deg[c] == 0) { if (!in_cover[c] && deg[c] > 0 && top.first != deg[c]) { heap.push_back({deg[c], c}); std::push_heap(heap.begin(), heap.end()); } continue; }
                in_cover[c] = 1;
                cover.push_back(c);
                for (int64_t k = ap[c]; k < ap[c + 1]; ++k) {
                    const int nb = an[k];
                    if (in_cover[nb]) continue;
                    deg[nb]--; alive--;
                }
                deg[c] = 0;
            }
            if (!cover.empty() && (int64_t)cover.size() * 8 <= n_cam) {
                n_hubs = (int)cover.size();
                int pos = 0;
                for (int64_t c = 0; c < n_cam; ++c) if (!in_cover[c]) pre[c] = pos++;
                std::sort(cover.begin(), cover.end());
                for (int c : cover) pre[c] = pos++;
                tile_weights(pre, acnt, adjm);   // the tile graph of the new camera order
            }
        }
    }
    // border = the tiles from the first hub camera on (at least the last tile, which may hold padding rows and stays
    // last = eliminated last, no fill)
    tr.mark("order: vertex cover");
    n_border_tiles = n_hubs > 0 ? nt - (int)((n_cam - n_hubs) / cpt) : 1;
    n_border_tiles = std::max(1, std::min(n_border_tiles, nt));
    std::vector<int> tperm(nt);
    std::iota(tperm.begin(), tperm.end(), 0);
    if (o.use_nd && nt - n_border_tiles >= 23) tperm = TilePlan::order(nt, adjm, true, o.nd_leaf, n_border_tiles);
    tr.mark("order: nested dissection");
    cmap.resize(n_cam);
    cinv.assign(n_cam, -1);
    for (int64_t c = 0; c < n_cam; ++c) {
        cmap[c] = (int)((int64_t)tperm[pre[c] / cpt] * cpt + pre[c] % cpt);
        cinv[cmap[c]] = (int)c;
    }
    present.assign((size_t)nt * nt, 0);
    for (int a = 0; a < nt; ++a)
        for (int b = 0; b < nt; ++b)
            if (acnt[(size_t)a * nt + b]) {
                const int I = std::max(tperm[a], tperm[b]), J = std::min(tperm[a], tperm[b]);
                present[(size_t)I * nt + J] = 1;
            }
    { std::vector<uint8_t>().swap(adjm); std::vector<uint32_t>().swap(acnt); }
    cam_i_.resize(n_obs);
    parallel_ranges(n_obs, 1 << 16, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) cam_i_[i] = (uint32_t)cmap[cam_idx[i]]; });
    n_present = 0;
    for (uint8_t b : present) n_present += b;
    tr.mark("order: maps");
    // ---- the plan's partition settings (host calls only) ------------------------------------------------------------------
    tp.set_partition(rank, (o.dist_factor && world > 1) ? world : 1);
    tp.set_own_all(false);
    if (o.dist_selftest > 1 && world == 1) {  // self-test: the distributed schedule for that many ranks, all played by this one
        tp.set_partition(0, o.dist_selftest);
        tp.set_own_all(true);
        TilePlan::Comm tc;
        tc.sum = [](double*, size_t, hipStream_t) { return true; };
        tc.max_int = [](int*, size_t, hipStream_t) { return true; };
        tp.set_comm(std::move(tc));
    }
    seconds[0] = now_s() - t_begin;
}

std::string BaHostStructure::build_obs_lists(const uint32_t* cam_idx, const uint32_t* pt_idx, const double* obs_uv,
                                             const BaStructOptions& o, TilePlan& tp) {
    SetupTrace tr;
    const int cpt = kNB / dc;
    const int rank = o.rank, world = o.world;
    std::vector<int64_t>& lp = lp_;
    raw_vector<int>& lobs = lobs_;
    (void)cam_idx;
    // ---- landmark sharding along the partition of the elimination tree ---------------------------------------------------------
    const double t1 = now_s();
    lmap.resize(n_pt);
    std::iota(lmap.begin(), lmap.end(), 0);
    tree_shard = false;
    lm_lo = 0; lm_hi = n_pt;
    if (false) tp.set_partition(rank, (o.dist_factor && world > 1) ? world : 1);
    pad_rank = 0;
    lam_mask.clear();
    // Tree sharding (distributed Cholesky, no communicator-less test shards): a landmark's cameras form a clique of
    // S, so their tile columns lie on ONE root path of the elimination tree -- below the shared top they all belong
    // to one rank.  Giving every landmark to that rank makes the tiles of a rank's own columns COMPLETE locally:
    // no reduce of S at all, only the top tiles are summed (which the distributed factorisation does anyway).
    // Landmarks seen by top cameras only go to the least loaded rank.  Landmarks are renumbered so that every
    // rank's set is one contiguous internal range.
    const std::vector<int> owner = needs_owner_preview(o) ? tp.preview_owners(nt, present) : std::vector<int>();
    if (!owner.empty()) {
        std::vector<int> lm_owner(n_pt, -1);
        std::vector<int64_t> load(world, 0);
        for (int64_t l = 0; l < n_pt; ++l) {
            for (int64_t k = lp[l]; k < lp[l + 1]; ++k) {
                const int ow = owner[cam_i_[lobs[k]] / cpt];
                if (ow >= 0) { lm_owner[l] = ow; break; }
            }
            if (lm_owner[l] >= 0) load[lm_owner[l]] += lp[l + 1] - lp[l];
        }
        for (int64_t l = 0; l < n_pt; ++l)
            if (lm_owner[l] < 0) {
                const int r = (int)(std::min_element(load.begin(), load.end()) - load.begin());
                lm_owner[l] = r;
                load[r] += lp[l + 1] - lp[l];
            }
        std::vector<int64_t> first(world + 1, 0);
        for (int64_t l = 0; l < n_pt; ++l) first[lm_owner[l] + 1]++;
        for (int r = 0; r < world; ++r) first[r + 1] += first[r];
        lm_lo = first[rank]; lm_hi = first[rank + 1];
        for (int64_t l = 0; l < n_pt; ++l) lmap[l] = (int)first[lm_owner[l]]++;
        tree_shard = true;
        // lambda on a camera's diagonal block: by the owner of its column, rank 0 for the shared top.  The identity on
        // the padding rows of the last tile follows the same rule: the elimination tree can be a forest (disconnected
        // covisibility), the last column is then a subtree root below the shared top, and its owner's copy of the
        // diagonal tile is the only one that is ever factorised.
        pad_rank = owner[nt - 1] >= 0 ? owner[nt - 1] : 0;
        lam_mask.resize(n_cam);
        for (int64_t ci = 0; ci < n_cam; ++ci) {
            const int ow = owner[ci / cpt];
            lam_mask[ci] = (ow == rank || (ow < 0 && rank == 0)) ? 1 : 0;
        }
    }

    // ---- landmark-major lists of the full problem in the INTERNAL landmark numbering --------------------------------------
    pt_i_.resize(n_obs);
    parallel_ranges(n_obs, 1 << 16, [&](int64_t b, int64_t e) { for (int64_t i = b; i < e; ++i) pt_i_[i] = (uint32_t)lmap[pt_idx[i]]; });
    full_ptr_.assign(n_pt + 1, 0);
    for (int64_t l = 0; l < n_pt; ++l) full_ptr_[lmap[l] + 1] = lp[l + 1] - lp[l];
    for (int64_t l = 0; l < n_pt; ++l) full_ptr_[l + 1] += full_ptr_[l];
    tr.mark("lists: sharding, pointers");
    full_obs_.resize(n_obs);
    parallel_ranges(n_pt, 8192, [&](int64_t b, int64_t e) {
        std::vector<std::pair<uint32_t, int>> tmp;
        for (int64_t l = b; l < e; ++l) {
            int* dst = full_obs_.data() + full_ptr_[lmap[l]];
            const int64_t k = lp[l + 1] - lp[l];
            // inside a landmark the observations are ordered by camera: the partners (cam_j <= cam_i) of an
            // observation are then a PREFIX of its landmark's list.  (camera, position) pairs: a stable order without
            // gathering the camera of an observation at every comparison
            tmp.resize((size_t)k);
            for (int64_t x = 0; x < k; ++x) { const int i = lobs[lp[l] + x]; tmp[x] = {cam_i_[i], i}; }
            bool ordered = true;
            for (int64_t x = 1; x < k; ++x) ordered = ordered && tmp[x - 1].first <= tmp[x].first;
            if (!ordered) std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<uint32_t, int>& a, const std::pair<uint32_t, int>& bb) { return a.first < bb.first; });
            for (int64_t x = 0; x < k; ++x) dst[x] = tmp[x].second;
        }
    });
    { raw_vector<int>().swap(lobs); std::vector<int64_t>().swap(lp); }
    tr.mark("lists: sort inside landmarks");
    // ---- shard: contiguous landmark range balanced by observation count (tree sharding: set above) --------
    if (!tree_shard) shard_range(n_pt, full_ptr_.data(), rank, world, &lm_lo, &lm_hi);
    const int64_t o_lo = full_ptr_[lm_lo], o_hi = full_ptr_[lm_hi];
    const int64_t n_loc = o_hi - o_lo;
    if (n_loc > 2000000000LL) return "too many observations on one rank";
    const bool dev_gather = o.device_gathers && world == 1;
    o_cam.resize(n_loc); o_pt.resize(n_loc); o_uv.resize(dev_gather ? 0 : 2 * n_loc); o_orig.resize(n_loc);
    pt_ptr.resize(n_pt + 1);
    for (int64_t l = 0; l <= n_pt; ++l) {
        int64_t p = full_ptr_[l];
        p = std::min(std::max(p, o_lo), o_hi) - o_lo;
        pt_ptr[l] = (int)p;
    }
    parallel_ranges(n_loc, 1 << 16, [&](int64_t b, int64_t e) {
        for (int64_t k = b; k < e; ++k) {
            const int i = full_obs_[o_lo + k];
            o_orig[k] = i;
            o_cam[k] = cam_i_[i]; o_pt[k] = pt_i_[i];
            if (!dev_gather) { o_uv[2 * k] = obs_uv[2 * (int64_t)i]; o_uv[2 * k + 1] = obs_uv[2 * (int64_t)i + 1]; }
        }
    });
    tr.mark("lists: landmark-major copies");
    // ---- camera-major lists over the local observations (+ copies of landmark and measurement) -----------------------------
    parallel_bucket_small(n_loc, n_cam, o_cam.data(), cam_ptr, cam_obs);
    co_pt.resize(dev_gather ? 0 : n_loc); co_uv.resize(dev_gather ? 0 : 2 * n_loc); co_rank.resize(n_loc);
    parallel_ranges(n_loc, 1 << 16, [&](int64_t b, int64_t e) {
        for (int64_t k = b; k < e; ++k) {
            const int i = cam_obs[k];
            if (!dev_gather) { co_pt[k] = o_pt[i]; co_uv[2 * k] = o_uv[2 * (size_t)i]; co_uv[2 * k + 1] = o_uv[2 * (size_t)i + 1]; }
            co_rank[k] = i - pt_ptr[o_pt[i]];
        }
    });
    n_pairs = 0;
    for (int64_t l = lm_lo; l < lm_hi; ++l) { const int64_t k = pt_ptr[l + 1] - pt_ptr[l]; n_pairs += k * (k + 1) / 2; }   // 4 M adds
    tr.mark("lists: camera-major");
    seconds[1] = now_s() - t1;
    return "";
}

void BaHostStructure::release_scratch() {
    raw_vector<uint32_t>().swap(cam_i_); raw_vector<uint32_t>().swap(pt_i_);
    std::vector<int64_t>().swap(full_ptr_); raw_vector<int>().swap(full_obs_);
}

void BaHostStructure::build_schur_lists(const BaStructOptions& o, const int* slot_host, PairDeviceTables* dev_tables) {
    const double t0 = now_s();
    pl = PairLists();
    if (o.schur_form >= 0)   // (< 0: a matrix-free-only handle -- S is never reduced, no lists)
        // every camera pair (i, j) of a landmark, sorted by the block S(cam_i, cam_j) it adds to; nine-column cameras in the
        // queued layout (form 4), whose records the device can write itself (dev_tables)
        build_pair_lists(dc, nt, slot_host, n_cam, cinv.data(), o_cam.data(), o_pt.data(), pt_ptr.data(), cam_ptr.data(),
                         cam_obs.data(), &pl, /*task_slots=*/0, /*queued=*/o.schur_form == 4 && dc == 9,
                         (o.schur_form == 4 && dc == 9) ? dev_tables : nullptr);
    seconds[3] = now_s() - t0;
}

}  // namespace apex
