// Scene description for the renderer: OBJ + MTL -> tables (mesh, BVH2, materials, lights).
//
// Restates the DECISIONS of the reference's scene compiler (src/driver/converter.cpp) as data
// instead of generated Impala source:
//   material clean-up (missing -> dummy, duplicates merged, unused dropped)   converter.cpp:467-557
//   BSDF choice from MTL: illum 5 mirror, illum 7 glass(1, Ni), else diffuse(Kd) / Phong(Ks, Ns)
//     mixed by luminance ratio; emissive iff Ke != 0                          converter.cpp:858-920
//   triangle lights with precomputed normal / inverse area                    converter.cpp:770-851
//   mesh buffers padded to float4 / int4 like the GPU targets                 converter.cpp:629-632,403-426
//   BVH2/Tri1 with geom_id = material id                                      converter.cpp:262-383,713-720
//   map_Kd / map_Ks textures (PNG, JPEG, TGA next to the OBJ) loaded into one RGBA8 pool        converter.cpp:595-610,749-768
// map_Ke (textured emitters): such lights use their constant Ke.  (The reference emits `make_triangle_light(math, v0, v1, v2,
// make_texture(...))` for it, converter.cpp:795-803, which passes a Texture where light.impala:140 takes a Color -- it cannot compile.)
#pragma once
#include <string>
#include <vector>
#include "../../include/rodent_render.h"
#include "mesh.h"

namespace rodent {

struct SceneData {
    std::vector<float>   vertices, normals, face_normals;   // float4 per element
    std::vector<int32_t> indices;                            // int4 per triangle
    std::vector<Node2>   nodes;
    std::vector<Tri1>    tris;
    std::vector<RodentMaterial> materials;
    std::vector<RodentLight>    lights;
    std::vector<int32_t> light_ids;
    std::vector<float>   texcoords;                          // float4 per vertex (u, v, 0, 0)
    std::vector<RodentTexture> textures;
    std::vector<uint32_t> texels;                            // RGBA8 pool
    std::vector<std::string> material_names, texture_names;
    int32_t default_spp = 4, default_max_path_len = 64;      // converter.cpp:1007-1012

    size_t num_tris() const { return indices.size() / 4; }
    RodentSceneDesc desc() const;
};

struct BuildParams;
// bvh: builder parameters other than the defaults (arity is always 2)
bool build_scene_from_obj(const std::string& obj_path, SceneData& scene, const BuildParams* bvh = nullptr);
bool save_scene(const std::string& path, const SceneData& scene);   // ".rscene" binary
// validates counts against the file size and every index (validate_scene)
bool load_scene(const std::string& path, SceneData& scene);
bool validate_scene(const SceneData& scene, std::string* why = nullptr);

// The reference converter's data directory (LZ4 buffer files, src/driver/buffer.h; converter.cpp:403-437,805-815,848):
// vertices / normals / face_normals / texcoords (float4 per element), indices (int4), bvh.bin (BVH2/Tri1 layout),
// light_ids, light_verts / light_norms / light_areas / light_colors.  Materials live in the reference's generated
// Impala source, not in these files.
bool save_reference_data(const std::string& dir, const SceneData& scene);
// Reads the same files back into the mesh / BVH / light fields of `scene` (materials are left alone).
bool load_reference_data(const std::string& dir, SceneData& scene);

} // namespace rodent
