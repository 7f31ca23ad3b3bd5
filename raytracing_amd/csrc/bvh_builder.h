// Host-side scene flattening and BVH construction.
//
// The tree must be IDENTICAL to the one the reference builds (node numbering, leaf contents
// and the order of triangles inside a leaf), because the nearest-hit tie-breaks of both
// traversals depend on it.  Reference: raytracer.cpp:897-1203 (BVH::BVH, BVHNode::Subdivide,
// SortX/Y/Z), raytracer.h:154-172 (Triangle constructor), raytracer.h:251-308 (AABB).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/mrt.h"
#include "mrt_math.h"

namespace mrt {

// One flattened triangle; index in the vector = the "hit ID" the API reports.
struct HostTri {
    f3 p0, p1, p2, centroid, normal;
    f2 uv0, uv1, uv2;
    float refl, refr, rindex;
    int32_t tex;
};

struct HostBVH {
    std::vector<HostTri> tris;
    std::vector<uint32_t> idx;          // BVH::m_TriangleIdx
    std::vector<f3> nmin, nmax;         // node bounds
    std::vector<uint32_t> first, count; // BVHNode::firstLeft / count (count == 0: inner node)
    uint32_t n_tris = 0, n_nodes_alloc = 0, pool_ptr = 0;
    int max_depth = 0;
};

// Mesh-by-mesh, triangle-by-triangle flattening (raytracer.cpp:908-929).  Returns false and
// sets err on malformed input (null arrays, vertex index out of range).
bool flatten_meshes(const mrt_mesh* meshes, int n_meshes, int n_textures, bool need_textures,
                    std::vector<HostTri>& out, std::string& err);

void build_bvh(HostBVH& bvh);

// lbvh.hip: a linear BVH built on the current GPU instead (not the reference's tree; see there).
bool build_lbvh_device(HostBVH& bvh, hipStream_t stream, std::string& err, float inner_cost = 0.0f);

}  // namespace mrt
