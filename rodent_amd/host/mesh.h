// Triangle mesh + OBJ/MTL loading for the scene tools.
// Behaviour follows the reference's loader (src/driver/obj.cpp:104-255 parser,
// :257-371 MTL, :412-509 compute_tri_mesh): material 0 is the unnamed default,
// faces are fan-triangulated, vertices are de-duplicated per OBJ object on the
// (v, t, n) index triple, the index buffer holds 4 x u32 per triangle
// (v0, v1, v2, material) and missing normals are rebuilt from face normals.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>
#include "vec.h"

namespace rodent {

struct Material {
    V3 ka, kd, ks, ke, tf;
    float ns = 1.0f, ni = 1.0f, tr = 0.0f, d = 1.0f;
    int illum = 2;
    std::string map_ka, map_kd, map_ks, map_ke, map_bump, map_d;
};

struct TriMesh {
    std::vector<V3>       vertices;
    std::vector<uint32_t> indices;        // 4 per triangle: v0 v1 v2 material
    std::vector<V3>       normals;        // per vertex
    std::vector<V3>       face_normals;   // per triangle
    std::vector<V2>       texcoords;      // per vertex
    std::vector<std::string> material_names;   // index = material id in `indices`
    std::vector<std::string> mtl_libs;

    size_t num_tris() const { return indices.size() / 4; }
    Triangle tri(size_t i) const {
        return {vertices[indices[4 * i]], vertices[indices[4 * i + 1]], vertices[indices[4 * i + 2]]};
    }
    std::vector<Triangle> triangles() const {
        std::vector<Triangle> t(num_tris());
        for (size_t i = 0; i < t.size(); i++) t[i] = tri(i);
        return t;
    }
};

// Returns false (after printing to stderr) when the file cannot be read or has
// malformed statements, like obj::load_obj.
bool load_obj(const std::string& path, TriMesh& mesh);
bool load_mtl(const std::string& path, std::unordered_map<std::string, Material>& lib);
bool save_obj(const std::string& path, const TriMesh& mesh);

} // namespace rodent
