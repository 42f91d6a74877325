#include "scene.h"

#include <algorithm>
#include <cstring>
#include <iostream>
#include <unordered_map>
#include <unordered_set>

#include "buffer_io.h"
#include "bvh_build.h"
#include "image.h"

namespace rodent {
namespace {

bool same(V3 a, V3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
bool zero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
float lum(V3 c) { return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f; }

bool same_material(const Material& a, const Material& b) {            // converter.cpp:440-459
    return same(a.ka, b.ka) && same(a.kd, b.kd) && same(a.ks, b.ks) && same(a.ke, b.ke) && a.ns == b.ns && a.ni == b.ni &&
           same(a.tf, b.tf) && a.illum == b.illum && a.map_kd == b.map_kd && a.map_ks == b.map_ks && a.map_ke == b.map_ke;
}

RodentMaterial to_table(const Material& m) {                           // converter.cpp:872-918
    RodentMaterial r; std::memset(&r, 0, sizeof r);
    r.kd[0] = m.kd.x; r.kd[1] = m.kd.y; r.kd[2] = m.kd.z;
    r.ks[0] = m.ks.x; r.ks[1] = m.ks.y; r.ks[2] = m.ks.z;
    r.tf[0] = m.tf.x; r.tf[1] = m.tf.y; r.tf[2] = m.tf.z;
    r.ns = m.ns; r.ni = m.ni;
    if (m.illum == 5) r.type = RODENT_BSDF_MIRROR;
    else if (m.illum == 7) r.type = RODENT_BSDF_GLASS;
    else {
        const bool diffuse = !zero(m.kd) || !m.map_kd.empty(), specular = !zero(m.ks) || !m.map_ks.empty();   // converter.cpp:875-876
        if (diffuse && specular) {
            const float ls = lum(m.ks), ld = lum(m.kd);
            r.type = RODENT_BSDF_MIX; r.mix_k = (ls + ld == 0.0f) ? 0.0f : ls / (ls + ld);
        } else if (specular) r.type = RODENT_BSDF_PHONG;
        else if (diffuse) r.type = RODENT_BSDF_DIFFUSE;
        else r.type = RODENT_BSDF_BLACK;
    }
    r.emissive = !zero(m.ke);
    return r;
}

} // namespace

RodentSceneDesc SceneData::desc() const {
    RodentSceneDesc d; std::memset(&d, 0, sizeof d);
    d.vertices = vertices.data(); d.normals = normals.data(); d.face_normals = face_normals.data(); d.indices = indices.data();
    d.nodes = nodes.data(); d.tris = tris.data(); d.materials = materials.data(); d.lights = lights.data(); d.light_ids = light_ids.data();
    d.num_vertices = (int32_t)(vertices.size() / 4); d.num_tris = (int32_t)num_tris(); d.num_nodes = (int32_t)nodes.size();
    d.num_bvh_tris = (int32_t)tris.size(); d.num_materials = (int32_t)materials.size(); d.num_lights = (int32_t)lights.size();
    d.texcoords = texcoords.data(); d.textures = textures.data(); d.texels = texels.data();
    d.num_textures = (int32_t)textures.size(); d.num_texels = (uint32_t)texels.size();
    return d;
}

bool build_scene_from_obj(const std::string& obj_path, SceneData& scene, const BuildParams* bvh_params) {
    TriMesh mesh;
    if (!load_obj(obj_path, mesh)) return false;
    std::unordered_map<std::string, Material> lib;
    for (auto& l : mesh.mtl_libs)
        if (!load_mtl(l, lib)) { std::cerr << "Invalid MTL file '" << l << "'" << std::endl; return false; }

    // dummy material for faces without (known) material (converter.cpp:469-486)
    Material dummy; dummy.kd = V3(0.0f, 1.0f, 1.0f); dummy.ns = 1.0f; dummy.ni = 1.0f; dummy.tr = 1.0f; dummy.d = 1.0f; dummy.illum = 2;
    lib[""] = dummy;
    std::vector<std::string> names = mesh.material_names;
    for (auto& n : names)
        if (!n.empty() && !lib.count(n)) {
            std::clog << "Missing material definition for '" << n << "'. Replaced by dummy material." << std::endl; n = ""; }

    if (mesh.num_tris() == 0) { std::cerr << "The OBJ file '" << obj_path << "' has no faces" << std::endl; return false; }
    // merge identical materials, drop unused ones (first occurrence keeps its place)
    const size_t nt = mesh.num_tris();
    std::vector<int> canon(names.size());
    for (size_t i = 0; i < names.size(); i++) {
        canon[i] = (int)i;
        for (size_t j = 0; j < i; j++) if (same_material(lib[names[i]], lib[names[j]])) { canon[i] = canon[j]; break; }
    }
    std::vector<int> used(names.size(), 0);
    for (size_t t = 0; t < nt; t++) used[canon[mesh.indices[4 * t + 3]]] = 1;
    std::vector<int> new_id(names.size(), -1);
    // textures: one pool entry per distinct file (converter.cpp:595-610); a file that cannot be decoded becomes the
    // reference's 1x1 black dummy image (converter.cpp:752,766) with a warning
    const size_t slash = obj_path.find_last_of("/\\");
    const std::string base = slash == std::string::npos ? std::string(".") : obj_path.substr(0, slash);
    std::unordered_map<std::string, int> tex_ids;
    auto texture_of = [&](std::string name) -> int {
        if (name.empty()) return 0;
        std::replace(name.begin(), name.end(), '\\', '/');                          // fix_file
        auto it = tex_ids.find(name);
        if (it != tex_ids.end()) return it->second;
        ImageRgba8 img; std::string err;
        if (!load_image(base + "/" + name, img, &err)) {
            std::clog << "Cannot load texture '" << name << "' (" << err << "). Replaced by a black image." << std::endl;
            img.width = img.height = 1; img.pixels.assign(4, 0);
        }
        RodentTexture t; t.width = img.width; t.height = img.height; t.offset = (uint32_t)scene.texels.size(); t.pad = 0;
        const size_t count = (size_t)img.width * img.height;
        scene.texels.resize(scene.texels.size() + count);
        std::memcpy(scene.texels.data() + t.offset, img.pixels.data(), count * 4);
        scene.textures.push_back(t); scene.texture_names.push_back(name);
        return tex_ids[name] = (int)scene.textures.size();                           // 1 + index
    };
    for (size_t i = 0; i < names.size(); i++)
        if (canon[i] == (int)i && used[i]) {
            new_id[i] = (int)scene.materials.size();
            RodentMaterial rm = to_table(lib[names[i]]);
            if (rm.type == RODENT_BSDF_DIFFUSE || rm.type == RODENT_BSDF_PHONG || rm.type == RODENT_BSDF_MIX) {
                rm.tex_kd = texture_of(lib[names[i]].map_kd); rm.tex_ks = texture_of(lib[names[i]].map_ks);
            }
            scene.materials.push_back(rm);
            scene.material_names.push_back(names[i]);
        }
    for (size_t t = 0; t < nt; t++) mesh.indices[4 * t + 3] = (uint32_t)new_id[canon[mesh.indices[4 * t + 3]]];

    // mesh buffers
    auto pad4 = [](const std::vector<V3>& v) { std::vector<float> o(v.size() * 4, 0.0f);
        for (size_t i = 0; i < v.size(); i++) { o[4 * i] = v[i].x; o[4 * i + 1] = v[i].y; o[4 * i + 2] = v[i].z; } return o; };
    scene.vertices = pad4(mesh.vertices); scene.normals = pad4(mesh.normals); scene.face_normals = pad4(mesh.face_normals);
    scene.indices.assign(mesh.indices.begin(), mesh.indices.end());
    scene.texcoords.assign(mesh.vertices.size() * 4, 0.0f);
    for (size_t i = 0; i < mesh.texcoords.size() && i < mesh.vertices.size(); i++) { scene.texcoords[4 * i] = mesh.texcoords[i].x;
        scene.texcoords[4 * i + 1] = mesh.texcoords[i].y; }

    // lights (converter.cpp:770-851): one per emissive triangle
    scene.light_ids.assign(nt, 0);
    for (size_t t = 0; t < nt; t++) {
        const int mid = scene.indices[4 * t + 3];
        if (!scene.materials[mid].emissive) continue;
        const Material& m = lib[scene.material_names[mid]];
        const Triangle tr = mesh.tri(t);
        V3 n = cross(tr.v1 - tr.v0, tr.v2 - tr.v0);
        const float inv_area = 1.0f / (0.5f * length(n));
        n = normalize(n);
        RodentLight L; std::memset(&L, 0, sizeof L);
        for (int k = 0; k < 3; k++) { L.v0[k] = tr.v0[k]; L.v1[k] = tr.v1[k]; L.v2[k] = tr.v2[k]; L.n[k] = n[k]; L.color[k] = m.ke[k]; }
        L.inv_area = inv_area;
        scene.light_ids[t] = (int32_t)scene.lights.size();
        scene.lights.push_back(L);
    }

    // BVH2/Tri1 with the material id as geometry id
    const std::vector<Triangle> tris = mesh.triangles();
    std::vector<uint32_t> geom(nt);
    for (size_t t = 0; t < nt; t++) geom[t] = (uint32_t)scene.indices[4 * t + 3];
    BuildParams p; if (bvh_params) p = *bvh_params;
    p.arity = 2;
    const WideBvh bvh = build_wide_bvh(tris, p);
    layout_bvh2_tri1(bvh, tris, geom.data(), scene.nodes, scene.tris);
    return true;
}

// ---- .rscene (version 2): magic, version, defaults, counts, then the arrays in declaration order ----
namespace { const uint32_t kMagic = 0x43534452u /* "RDSC" */, kVersion = 2; }

bool save_scene(const std::string& path, const SceneData& s) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const uint32_t hdr[12] = {kMagic, kVersion, (uint32_t)s.default_spp, (uint32_t)s.default_max_path_len,
                              (uint32_t)(s.vertices.size() / 4), (uint32_t)s.num_tris(), (uint32_t)s.nodes.size(),
                              (uint32_t)s.tris.size(), (uint32_t)s.materials.size(), (uint32_t)s.lights.size(),
                              (uint32_t)s.textures.size(), (uint32_t)s.texels.size()};
    bool ok = fwrite(hdr, 4, 12, f) == 12;
    auto put = [&](const void* p, size_t bytes) { ok = ok && (bytes == 0 || fwrite(p, 1, bytes, f) == bytes); };
    put(s.vertices.data(), s.vertices.size() * 4); put(s.normals.data(), s.normals.size() * 4);
    put(s.face_normals.data(), s.face_normals.size() * 4); put(s.indices.data(), s.indices.size() * 4);
    put(s.nodes.data(), s.nodes.size() * sizeof(Node2)); put(s.tris.data(), s.tris.size() * sizeof(Tri1));
    put(s.materials.data(), s.materials.size() * sizeof(RodentMaterial)); put(s.lights.data(), s.lights.size() * sizeof(RodentLight));
    put(s.light_ids.data(), s.light_ids.size() * 4);
    put(s.texcoords.data(), s.texcoords.size() * 4); put(s.textures.data(), s.textures.size() * sizeof(RodentTexture));
    put(s.texels.data(), s.texels.size() * 4);
    fclose(f);
    return ok;
}

// Every index a kernel will follow is range-checked here, once: BVH child ids, leaf extents, geometry / vertex / light /
// texture ids.  (A corrupt file must fail at load, not as an out-of-bounds read in k_shade.)
bool validate_scene(const SceneData& s, std::string* why) {
    auto bad = [&](const char* m) { if (why) *why = m; return false; };
    const size_t nv = s.vertices.size() / 4, nt = s.indices.size() / 4;
    if (s.vertices.size() % 4 || s.indices.size() % 4 || s.normals.size() != s.vertices.size() || s.face_normals.size() != 4 * nt ||
        s.light_ids.size() != nt || (!s.texcoords.empty() && s.texcoords.size() != s.vertices.size())) return bad("table sizes disagree");
    if (s.nodes.empty() || s.tris.empty() || s.materials.empty()) return bad("empty BVH or material table");
    for (size_t t = 0; t < nt; t++) {
        for (int k = 0; k < 3; k++) if ((uint32_t)s.indices[4 * t + k] >= nv) return bad("vertex index out of range");
        if ((uint32_t)s.indices[4 * t + 3] >= s.materials.size()) return bad("material index out of range");
        if (s.light_ids[t] < 0 || (s.light_ids[t] > 0 && (size_t)s.light_ids[t] >= s.lights.size())) return bad("light id out of range");
        // the shader looks an emitter's triangle up in the light table (light id 0 included): the entry must exist
        if (s.materials[s.indices[4 * t + 3]].emissive
            && (size_t)s.light_ids[t] >= s.lights.size()) return bad("emissive triangle without an entry in the light table");
    }
    for (const Node2& n : s.nodes)
        for (int k = 0; k < 2; k++) {
            const int32_t c = n.child[k];
            if (c > 0 && (size_t)c > s.nodes.size()) return bad("BVH child id out of range");
            if (c < 0 && (size_t)~c >= s.tris.size()) return bad("BVH leaf id out of range");
        }
    if (s.tris.back().prim_id >= 0) return bad("last BVH triangle lacks the end-of-leaf bit");
    for (const Tri1& t : s.tris) {
        if ((size_t)(t.prim_id & 0x7FFFFFFF) >= nt) return bad("BVH triangle refers to a primitive that does not exist");
        if ((uint32_t)t.geom_id >= s.materials.size()) return bad("BVH triangle refers to a material that does not exist");
    }
    for (const RodentMaterial& m : s.materials)
        if (m.tex_kd < 0 || (size_t)m.tex_kd > s.textures.size() || m.tex_ks < 0
            || (size_t)m.tex_ks > s.textures.size()) return bad("material refers to a texture that does not exist");
    for (const RodentTexture& t : s.textures)
        if (t.width <= 0 || t.height <= 0
            || (uint64_t)t.offset + (uint64_t)t.width * (uint64_t)t.height > s.texels.size()) return bad("texture outside the texel pool");
    return true;
}

bool load_scene(const std::string& path, SceneData& s) {
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    uint32_t hdr[12];
    bool ok = fread(hdr, 4, 12, f) == 12 && hdr[0] == kMagic && hdr[1] == kVersion;
    if (ok) {
        // the header's counts must add up to the file's size before anything is allocated
        const uint64_t nv = hdr[4], nt = hdr[5];
        const uint64_t expect = 48 + 16 * nv * 2 + 16 * nt * 2 + sizeof(Node2) * (uint64_t)hdr[6] + sizeof(Tri1) * (uint64_t)hdr[7]
            + sizeof(RodentMaterial) * (uint64_t)hdr[8] +
                                sizeof(RodentLight) * (uint64_t)hdr[9] + 4 * nt + 16 * nv + sizeof(RodentTexture) * (uint64_t)hdr[10]
                                    + 4ull * hdr[11];
        ok = fseek(f, 0, SEEK_END) == 0 && (uint64_t)ftell(f) == expect && fseek(f, 48, SEEK_SET) == 0;
    }
    if (ok) {
        s.default_spp = (int32_t)hdr[2]; s.default_max_path_len = (int32_t)hdr[3];
        auto get = [&](auto& vec, size_t count) { vec.resize(count);
            ok = ok && (count == 0 || fread(vec.data(), sizeof(vec[0]), count, f) == count); };
        get(s.vertices, 4ull * hdr[4]); get(s.normals, 4ull * hdr[4]); get(s.face_normals, 4ull * hdr[5]); get(s.indices, 4ull * hdr[5]);
        get(s.nodes, hdr[6]); get(s.tris, hdr[7]); get(s.materials, hdr[8]); get(s.lights, hdr[9]); get(s.light_ids, hdr[5]);
        get(s.texcoords, 4ull * hdr[4]); get(s.textures, hdr[10]); get(s.texels, hdr[11]);
    }
    fclose(f);
    std::string why;
    if (ok && !validate_scene(s, &why)) { std::cerr << "Invalid scene file '" << path << "': " << why << std::endl; ok = false; }
    return ok;
}

bool save_reference_data(const std::string& dir, const SceneData& s) {
    std::remove((dir + "/bvh.bin").c_str());                              // bvh.bin is appended to (converter.cpp:430,716)
    std::vector<float> light_verts, light_norms, light_areas, light_colors;
    for (const RodentLight& L : s.lights) {
        for (const float* v : {L.v0, L.v1, L.v2}) light_verts.insert(light_verts.end(), {v[0], v[1], v[2], 0.0f});
        light_norms.insert(light_norms.end(), {L.n[0], L.n[1], L.n[2], 0.0f});
        light_areas.push_back(L.inv_area);
        light_colors.insert(light_colors.end(), {L.color[0], L.color[1], L.color[2], 0.0f});
    }
    return write_buffer_file(dir + "/vertices.bin", s.vertices) && write_buffer_file(dir + "/normals.bin", s.normals) &&
           write_buffer_file(dir + "/face_normals.bin", s.face_normals) && write_buffer_file(dir + "/indices.bin", s.indices) &&
           write_buffer_file(dir + "/texcoords.bin", s.texcoords) && append_bvh_bin(dir + "/bvh.bin", s.nodes, s.tris) &&
           write_buffer_file(dir + "/light_ids.bin", s.light_ids) && write_buffer_file(dir + "/light_verts.bin", light_verts) &&
           write_buffer_file(dir + "/light_norms.bin", light_norms) && write_buffer_file(dir + "/light_areas.bin", light_areas) &&
           write_buffer_file(dir + "/light_colors.bin", light_colors);
}

bool load_reference_data(const std::string& dir, SceneData& s) {
    std::vector<float> light_verts, light_norms, light_areas, light_colors;
    if (!(read_buffer_file(dir + "/vertices.bin", s.vertices) && read_buffer_file(dir + "/normals.bin", s.normals) &&
          read_buffer_file(dir + "/face_normals.bin", s.face_normals) && read_buffer_file(dir + "/indices.bin", s.indices) &&
          read_buffer_file(dir + "/texcoords.bin", s.texcoords) && load_bvh_bin(dir + "/bvh.bin", s.nodes, s.tris) &&
          read_buffer_file(dir + "/light_ids.bin", s.light_ids) && read_buffer_file(dir + "/light_verts.bin", light_verts) &&
          read_buffer_file(dir + "/light_norms.bin", light_norms) && read_buffer_file(dir + "/light_areas.bin", light_areas) &&
          read_buffer_file(dir + "/light_colors.bin", light_colors)))
        return false;
    const size_t nl = light_areas.size();
    if (light_verts.size() != 12 * nl || light_norms.size() != 4 * nl || light_colors.size() != 4 * nl) return false;
    s.lights.assign(nl, RodentLight{});
    for (size_t i = 0; i < nl; i++) {
        RodentLight& L = s.lights[i];
        for (int k = 0; k < 3; k++) {
            L.v0[k] = light_verts[12 * i + k]; L.v1[k] = light_verts[12 * i + 4 + k]; L.v2[k] = light_verts[12 * i + 8 + k];
            L.n[k] = light_norms[4 * i + k]; L.color[k] = light_colors[4 * i + k];
        }
        L.inv_area = light_areas[i];
    }
    return true;
}

} // namespace rodent
