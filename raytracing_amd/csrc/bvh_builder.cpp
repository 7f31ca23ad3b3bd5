#include "bvh_builder.h"

#include "host_parallel.h"

#include <stdlib.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <thread>

namespace mrt {

bool flatten_meshes(const mrt_mesh* meshes, int n_meshes, int n_textures, bool need_textures,
                    std::vector<HostTri>& out, std::string& err) {
    out.clear();
    if (n_meshes < 0 || (n_meshes > 0 && !meshes)) { err = "meshes is null"; return false; }
    size_t total = 0;
    for (int m = 0; m < n_meshes; ++m) {
        const mrt_mesh& M = meshes[m];
        if (M.tris < 0 || M.verts < 0) { err = "negative vertex/triangle count"; return false; }
        if (M.tris > 0 && (!M.pos || !M.uv || !M.tri_normal || !M.tri)) { err = "mesh array is null"; return false; }
        if (need_textures && M.tris > 0 && (M.texture_id < 0 || M.texture_id >= n_textures)) {
            err = "mesh texture_id out of range (the tracer needs a texture on every mesh)";
            return false;
        }
        total += (size_t)M.tris;
    }
    if (total > 0x3fffffffu) { err = "too many triangles"; return false; }
    out.resize(total);
    size_t base = 0;
    for (int m = 0; m < n_meshes; ++m) {
        const mrt_mesh& M = meshes[m];
        const f3* pos = reinterpret_cast<const f3*>(M.pos);
        const f2* uv = reinterpret_cast<const f2*>(M.uv);
        const f3* N = reinterpret_cast<const f3*>(M.tri_normal);
        std::atomic<bool> bad{false};
        parallel_for((size_t)M.tris, [&](size_t jb, size_t je) {
            for (size_t j = jb; j < je; ++j) {
                const int32_t a = M.tri[j * 3], b = M.tri[j * 3 + 1], c = M.tri[j * 3 + 2];
                if (a < 0 || b < 0 || c < 0 || a >= M.verts || b >= M.verts || c >= M.verts) { bad = true; return; }
                HostTri& t = out[base + j];
                t.normal = N[j];
                t.p0 = pos[a]; t.p1 = pos[b]; t.p2 = pos[c];
                t.centroid = ((t.p0 + t.p1) + t.p2) / 3.0f;        // raytracer.h:164
                t.uv0 = uv[a]; t.uv1 = uv[b]; t.uv2 = uv[c];
                t.refl = M.refl; t.refr = M.refr; t.rindex = M.rindex;
                t.tex = M.texture_id;
            }
        });
        if (bad) { err = "triangle vertex index out of range"; out.clear(); return false; }
        base += (size_t)M.tris;
    }
    return true;
}

namespace {

struct KeyIdx { float key; uint32_t idx; };
struct Box { float lo[3], hi[3]; };
struct Span { int s, e; };

// Surface-area measure of raytracer.h:303-307: (dz*dx + dz*dy + dx*dy) * 2
inline float area(const Box& b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return (dz * dx + dz * dy + dx * dy) * 2;
}

class Builder {
public:
    explicit Builder(HostBVH& b) : B(b) {
        const size_t n = B.tris.size();
        for (int a = 0; a < 3; ++a) { cen[a].resize(n); tlo[a].resize(n); thi[a].resize(n); }
        for (size_t i = 0; i < n; ++i) {
            const HostTri& t = B.tris[i];
            const float p[3][3] = {{t.p0.x, t.p0.y, t.p0.z}, {t.p1.x, t.p1.y, t.p1.z}, {t.p2.x, t.p2.y, t.p2.z}};
            const float c[3] = {t.centroid.x, t.centroid.y, t.centroid.z};
            for (int a = 0; a < 3; ++a) {
                cen[a][i] = c[a];
                tlo[a][i] = gmin(p[0][a], gmin(p[1][a], p[2][a]));   // raytracer.h:286-291 inner terms
                thi[a][i] = gmax(p[0][a], gmax(p[1][a], p[2][a]));
            }
        }
        scratch.resize(n);
    }

    // Bounds of idx[start, start+count) accumulated front to back as raytracer.h:278-293 does.
    Box bounds_forward(uint32_t start, uint32_t count) const {
        Box b;
        for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
        for (uint32_t i = start; i < start + count; ++i) extend_forward(b, B.idx[i]);
        return b;
    }

    // The reference's recursive Hoare quicksort (raytracer.cpp:767-874) on centroid[axis].  Its
    // outer loop stops at L == R without looking at that element, which then belongs to BOTH
    // sub-ranges; the permutation among equal keys therefore depends on sorting [start,R]
    // completely before [L,end].  The work list is LIFO, so the right range is pushed first.
    void sort_axis(int axis, int start, int end, std::vector<Span>& work) {
        const float* key = cen[axis].data();
        for (int i = start; i <= end; ++i) scratch[i] = {key[B.idx[i]], B.idx[i]};
        work.clear();
        work.push_back({start, end});
        while (!work.empty()) {
            const auto [s, e] = work.back();
            work.pop_back();
            int L = s, R = e;
            const float center = scratch[(L + R) / 2].key;
            while (L < R) {
                while (scratch[L].key < center) L++;
                while (scratch[R].key > center) R--;
                if (L <= R) { std::swap(scratch[L], scratch[R]); L++; R--; }
            }
            if (L < e) work.push_back({L, e});
            if (s < R) work.push_back({s, R});
        }
        for (int i = start; i <= end; ++i) B.idx[i] = scratch[i].idx;
    }

    // One axis block of BVHNode::Subdivide (raytracer.cpp:984-1043 and its Y/Z twins): eight
    // candidate planes at min + (2+i)*extent/10, the plane position accumulated by repeated
    // addition.  After sort_axis the range is ordered, so "first triangle whose centroid is
    // beyond the plane" advances monotonically and the left/right bounds of every candidate are
    // one forward and one backward accumulation.
    void sweep_axis(int axis, uint32_t first, uint32_t count, float lo_axis, float extent,
                    float& best_cost, int& best_axis, int& split_out, std::vector<Span>& work) {
        sort_axis(axis, (int)first, (int)(first + count - 1), work);
        const float step = extent / 10.0f;
        float plane = step + lo_axis;
        plane += step;
        int cand[8];
        uint32_t k = 0;                       // offset inside the range
        for (int i = 0; i < 8; ++i) {
            while (k < count && !(scratch[first + k].key > plane)) ++k;
            cand[i] = (k < count) ? (int)k : -1;
            plane += step;
        }
        // left boxes: prefix accumulation up to the largest candidate
        Box left[8], right[8];
        {
            Box b;
            for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
            uint32_t done = 0;
            for (int i = 0; i < 8; ++i) {
                if (cand[i] < 0) break;
                for (; done < (uint32_t)cand[i]; ++done) extend_forward(b, B.idx[first + done]);
                left[i] = b;
            }
        }
        // right boxes: the reference accumulates [cand, end) front to back; walking backwards
        // with "<=" / ">=" keeps the same element on ties, so even the sign of a zero matches
        {
            Box b;
            for (int a = 0; a < 3; ++a) { b.lo[a] = INFINITY; b.hi[a] = -INFINITY; }
            uint32_t done = count;            // elements [done, count) are in b
            for (int i = 7; i >= 0; --i) {
                if (cand[i] < 0) continue;
                for (; done > (uint32_t)cand[i]; --done) extend_backward(b, B.idx[first + done - 1]);
                right[i] = b;
            }
        }
        int best_split = (int)first;
        for (int i = 0; i < 8; ++i) {
            if (cand[i] < 0) break;           // planes only move outwards: later ones find nothing either
            const int left_count = cand[i];
            const int right_count = (int)(count - (uint32_t)left_count);
            const float left_area = left_count == 0 ? 0.0f : area(left[i]);
            const float right_area = right_count == 0 ? 0.0f : area(right[i]);
            const float cost = left_area * left_count + right_area * right_count;
            if (cost < best_cost) {
                best_cost = cost;
                best_split = (int)first + cand[i] - 1;   // one before the evaluated partition (raytracer.cpp:1033)
                best_axis = axis;
            }
        }
        split_out = best_split;
    }

    // The tree is built as linked nodes first and numbered afterwards.  A node's split only touches its
    // own range of the index array, so the two children of a split are independent pieces of work: large
    // ones are handed to a pool of threads, small ones are finished by the thread that made them.  The
    // reference numbers nodes in the order its recursion allocates them (children pair first, then the
    // whole left subtree, then the right one; raytracer.cpp:1181-1199), which number_nodes reproduces,
    // so the arrays are the same whatever the threads did in which order.
    void run() {
        B.n_tris = (uint32_t)B.tris.size();
        B.n_nodes_alloc = B.n_tris * 2 + 1;
        const size_t slots = (size_t)B.n_nodes_alloc + 3;
        B.nmin.assign(slots, f3{0, 0, 0});
        B.nmax.assign(slots, f3{0, 0, 0});
        B.first.assign(slots, 0);
        B.count.assign(slots, 0);
        B.idx.resize(B.n_tris);
        for (uint32_t i = 0; i < B.n_tris; ++i) B.idx[i] = i;
        B.pool_ptr = 2;                       // raytracer.cpp:938: slots 1 and 2 stay unused
        B.max_depth = 0;
        B.first[0] = 0;
        B.count[0] = B.n_tris;
        if (B.n_tris == 0) return;

        int n_threads = 1;
        if (B.n_tris >= 4 * kTaskMin) {
            n_threads = (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
            if (const char* e = getenv("MRT_BUILD_THREADS")) n_threads = std::max(1, std::min(64, atoi(e)));
        }
        std::vector<std::unique_ptr<Worker>> workers;
        for (int i = 0; i < n_threads; ++i) workers.emplace_back(new Worker());
        TNode root;
        root.first = 0; root.count = B.n_tris; root.depth = 0;
        pending = 1;
        queue.push_back(&root);
        std::vector<std::thread> threads;
        for (int i = 1; i < n_threads; ++i) threads.emplace_back([this, &workers, i] { work_loop(*workers[i]); });
        work_loop(*workers[0]);
        for (std::thread& t : threads) t.join();

        // number the nodes in the reference's allocation order and write the arrays
        std::vector<std::pair<const TNode*, uint32_t>> st;
        st.push_back({&root, 0u});
        while (!st.empty()) {
            const auto [n, slot] = st.back();
            st.pop_back();
            B.max_depth = std::max(B.max_depth, n->depth);
            B.nmin[slot] = {n->box.lo[0], n->box.lo[1], n->box.lo[2]};
            B.nmax[slot] = {n->box.hi[0], n->box.hi[1], n->box.hi[2]};
            if (n->left) {
                B.pool_ptr += 2;
                const uint32_t l = B.pool_ptr - 1, r = B.pool_ptr;
                B.first[slot] = l;
                B.count[slot] = 0;
                st.push_back({n->right, r});      // the left subtree is numbered completely before the right one
                st.push_back({n->left, l});
            } else {
                B.first[slot] = n->first;
                B.count[slot] = n->count;
            }
        }
    }

private:
    static constexpr uint32_t kTaskMin = 8192;     // ranges at least this long are offered to other threads

    struct TNode {
        Box box;
        uint32_t first = 0, count = 0;
        int depth = 0;
        TNode* left = nullptr;
        TNode* right = nullptr;
    };
    struct Worker {
        std::deque<TNode> arena;                   // deque: growing never moves a node
        std::vector<Span> sort_work;
        std::vector<TNode*> local;
    };

    void work_loop(Worker& w) {
        for (;;) {
            TNode* n = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [this] { return !queue.empty() || pending == 0; });
                if (queue.empty()) return;          // pending == 0: the tree is complete
                n = queue.back();
                queue.pop_back();
            }
            // finish this node and everything below it that is too small to share
            w.local.clear();
            w.local.push_back(n);
            while (!w.local.empty()) {
                TNode* cur = w.local.back();
                w.local.pop_back();
                if (!split_node(*cur, w)) continue;
                for (TNode* c : {cur->right, cur->left}) {
                    if (c->count >= kTaskMin) {
                        std::lock_guard<std::mutex> lk(mu);
                        ++pending;
                        queue.push_back(c);
                        cv.notify_one();
                    } else {
                        w.local.push_back(c);
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--pending == 0) cv.notify_all();
            }
        }
    }

    bool split_node(TNode& self, Worker& w) {
        const uint32_t first = self.first, count = self.count;
        const Box nb = bounds_forward(first, count);
        self.box = nb;
        if (count < 3) return false;          // primPerNode (raytracer.h:353)
        const float init_cost = area(nb) * count;
        float best_cost = init_cost;
        int best_axis = 0;
        int split[3] = {0, 0, 0};
        for (int axis = 0; axis < 3; ++axis)
            sweep_axis(axis, first, count, nb.lo[axis], nb.hi[axis] - nb.lo[axis], best_cost, best_axis, split[axis], w.sort_work);
        sort_axis(best_axis, (int)first, (int)(first + count - 1), w.sort_work);     // raytracer.cpp:1165-1179
        if (init_cost == best_cost) return false;
        const uint32_t left_count = (uint32_t)(split[best_axis] - (int)first);
        const uint32_t right_count = count - left_count;
        if (left_count == 0 || right_count == 0) return false;
        w.arena.emplace_back();
        TNode* l = &w.arena.back();
        w.arena.emplace_back();
        TNode* r = &w.arena.back();
        l->first = first;                        l->count = left_count;  l->depth = self.depth + 1;
        r->first = (uint32_t)split[best_axis];   r->count = right_count; r->depth = self.depth + 1;
        self.left = l;
        self.right = r;
        return true;
    }

    inline void extend_forward(Box& b, uint32_t t) const {
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = gmin(tlo[a][t], b.lo[a]);
            b.hi[a] = gmax(thi[a][t], b.hi[a]);
        }
    }
    inline void extend_backward(Box& b, uint32_t t) const {
        for (int a = 0; a < 3; ++a) {
            b.lo[a] = (tlo[a][t] <= b.lo[a]) ? tlo[a][t] : b.lo[a];
            b.hi[a] = (thi[a][t] >= b.hi[a]) ? thi[a][t] : b.hi[a];
        }
    }

    HostBVH& B;
    std::vector<float> cen[3], tlo[3], thi[3];
    std::vector<KeyIdx> scratch;               // indexed like B.idx: concurrent nodes use disjoint parts
    std::mutex mu;
    std::condition_variable cv;
    std::vector<TNode*> queue;                 // nodes waiting for a thread
    size_t pending = 0;                        // queued + being worked on
};

}  // namespace

void build_bvh(HostBVH& bvh) {
    Builder b(bvh);
    b.run();
}

}  // namespace mrt
