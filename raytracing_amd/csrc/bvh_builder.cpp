#include "bvh_builder.h"

#include <algorithm>

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
    out.reserve(total);
    for (int m = 0; m < n_meshes; ++m) {
        const mrt_mesh& M = meshes[m];
        const f3* pos = reinterpret_cast<const f3*>(M.pos);
        const f2* uv = reinterpret_cast<const f2*>(M.uv);
        const f3* N = reinterpret_cast<const f3*>(M.tri_normal);
        for (int j = 0; j < M.tris; ++j) {
            const int32_t a = M.tri[j * 3], b = M.tri[j * 3 + 1], c = M.tri[j * 3 + 2];
            if (a < 0 || b < 0 || c < 0 || a >= M.verts || b >= M.verts || c >= M.verts) {
                err = "triangle vertex index out of range";
                return false;
            }
            HostTri t;
            t.normal = N[j];
            t.p0 = pos[a]; t.p1 = pos[b]; t.p2 = pos[c];
            t.centroid = ((t.p0 + t.p1) + t.p2) / 3.0f;        // raytracer.h:164
            t.uv0 = uv[a]; t.uv1 = uv[b]; t.uv2 = uv[c];
            t.refl = M.refl; t.refr = M.refr; t.rindex = M.rindex;
            t.tex = M.texture_id;
            out.push_back(t);
        }
    }
    return true;
}

namespace {

struct KeyIdx { float key; uint32_t idx; };
struct Box { float lo[3], hi[3]; };

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
    void sort_axis(int axis, int start, int end) {
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
                    float& best_cost, int& best_axis, int& split_out) {
        sort_axis(axis, (int)first, (int)(first + count - 1));
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
        struct Todo { uint32_t node; int depth; };
        std::vector<Todo> todo;
        todo.push_back({0, 0});
        while (!todo.empty()) {
            const Todo cur = todo.back();
            todo.pop_back();
            uint32_t left, right;
            if (split_node(cur.node, cur.depth, left, right)) {
                todo.push_back({right, cur.depth + 1});   // left subtree is finished first, as the recursion does
                todo.push_back({left, cur.depth + 1});
            }
        }
    }

private:
    bool split_node(uint32_t self, int depth, uint32_t& left, uint32_t& right) {
        B.max_depth = std::max(B.max_depth, depth);
        const uint32_t first = B.first[self], count = B.count[self];
        const Box nb = bounds_forward(first, count);
        B.nmin[self] = {nb.lo[0], nb.lo[1], nb.lo[2]};
        B.nmax[self] = {nb.hi[0], nb.hi[1], nb.hi[2]};
        if (count < 3) return false;          // primPerNode (raytracer.h:353)
        B.pool_ptr += 2;
        left = B.pool_ptr - 1;
        right = B.pool_ptr;
        const float init_cost = area(nb) * count;
        float best_cost = init_cost;
        int best_axis = 0;
        int split[3] = {0, 0, 0};
        for (int axis = 0; axis < 3; ++axis)
            sweep_axis(axis, first, count, nb.lo[axis], nb.hi[axis] - nb.lo[axis], best_cost, best_axis, split[axis]);
        sort_axis(best_axis, (int)first, (int)(first + count - 1));     // raytracer.cpp:1165-1179
        if (init_cost == best_cost) { B.pool_ptr -= 2; return false; }
        const uint32_t left_count = (uint32_t)(split[best_axis] - (int)first);
        const uint32_t right_count = count - left_count;
        if (left_count == 0 || right_count == 0) { B.pool_ptr -= 2; return false; }
        B.first[left] = first;                   B.count[left] = left_count;
        B.first[right] = (uint32_t)split[best_axis]; B.count[right] = right_count;
        B.first[self] = left;
        B.count[self] = 0;
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
    std::vector<KeyIdx> scratch;
    struct Span { int s, e; };
    std::vector<Span> work;
};

}  // namespace

void build_bvh(HostBVH& bvh) {
    Builder b(bvh);
    b.run();
}

}  // namespace mrt
