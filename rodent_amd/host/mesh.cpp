#include "mesh.h"

#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <limits>
#include <sstream>

namespace rodent {
namespace {

struct Corner { int v = 0, t = 0, n = 0; };
struct CornerHash {
    size_t operator()(const Corner& c) const {
        uint64_t h = 1469598103934665603ull;
        for (int x : {c.v, c.t, c.n}) { h ^= (uint32_t)x; h *= 1099511628211ull; }
        return (size_t)h;
    }
};
struct CornerEq { bool operator()(const Corner& a, const Corner& b) const { return a.v == b.v && a.t == b.t && a.n == b.n; } };

struct Face { std::vector<Corner> corners; int material; };
struct Object { std::vector<Face> faces; };   // groups only partition faces; order is what matters

// "12", "12/3", "12//5", "12/3/5"; negative values are relative (obj.cpp:68-100)
bool parse_corner(const std::string& tok, Corner& c) {
    if (tok.empty() || !(isdigit((unsigned char)tok[0]) || tok[0] == '-')) return false;
    const char* s = tok.c_str(); char* end;
    c.v = (int)strtol(s, &end, 10);
    if (*end == '/') {
        end++;
        if (*end != '/') c.t = (int)strtol(end, &end, 10);
        if (*end == '/') { end++; c.n = (int)strtol(end, &end, 10); }
    }
    return true;
}

std::string dirname_of(const std::string& p) {
    auto k = p.find_last_of("/\\");
    return k == std::string::npos ? std::string(".") : p.substr(0, k);
}

} // namespace

bool load_obj(const std::string& path, TriMesh& mesh) {
    std::ifstream in(path);
    if (!in) { std::cerr << "Cannot open OBJ file '" << path << "'" << std::endl; return false; }

    std::vector<V3> pos(1), nrm(1); std::vector<V2> tex(1);       // index 0 = dummy (obj.cpp:116-119)
    std::vector<Object> objects(1);
    std::vector<std::string> materials(1, "");
    int cur_mtl = 0, errors = 0, line_no = 0;
    std::string line;
    while (std::getline(in, line)) {
        line_no++;
        std::istringstream ls(line);
        std::string cmd;
        if (!(ls >> cmd) || cmd[0] == '#') continue;
        if (cmd == "v")       { V3 p; ls >> p.x >> p.y >> p.z; pos.push_back(p); }
        else if (cmd == "vn") { V3 p; ls >> p.x >> p.y >> p.z; nrm.push_back(p); }
        else if (cmd == "vt") { V2 p; ls >> p.x >> p.y; tex.push_back(p); }
        else if (cmd == "f") {
            Face f; f.material = cur_mtl;
            std::string tok; Corner c;
            while (ls >> tok) { c = Corner(); if (!parse_corner(tok, c)) break; f.corners.push_back(c); }
            if (f.corners.size() < 3) { std::cerr << "Invalid face (line " << line_no << ")." << std::endl; errors++; continue; }
            bool ok = true;
            for (auto& k : f.corners) {
                if (k.v < 0) k.v += (int)pos.size();
                if (k.t < 0) k.t += (int)tex.size();
                if (k.n < 0) k.n += (int)nrm.size();
                ok &= k.v > 0 && k.t >= 0 && k.n >= 0 && k.v < (int)pos.size() && k.t < (int)tex.size() && k.n < (int)nrm.size();
            }
            if (!ok) { std::cerr << "Invalid indices in face definition (line " << line_no << ")." << std::endl; errors++; continue; }
            objects.back().faces.push_back(std::move(f));
        }
        else if (cmd == "g" || cmd == "s") { /* groups keep face order; smoothing ignored */ }
        else if (cmd == "o") objects.emplace_back();
        else if (cmd == "usemtl") {
            std::string name; ls >> name;
            auto it = std::find(materials.begin(), materials.end(), name);
            cur_mtl = (int)(it - materials.begin());
            if (it == materials.end()) materials.push_back(name);
        }
        else if (cmd == "mtllib") { std::string name; ls >> name; mesh.mtl_libs.push_back(dirname_of(path) + "/" + name); }
        else { std::cerr << "Unknown command '" << cmd << "' (line " << line_no << ")." << std::endl; errors++; }
    }
    if (errors) return false;

    mesh.material_names = materials;
    for (auto& obj : objects) {
        std::unordered_map<Corner, uint32_t, CornerHash, CornerEq> remap;
        std::vector<Corner> order;
        std::vector<uint32_t> tri_idx;   // 4 per triangle, object-local vertex ids
        bool has_n = false, has_t = false;
        for (auto& f : obj.faces) {
            std::vector<uint32_t> ids;
            for (auto& c : f.corners) {
                auto it = remap.find(c);
                if (it == remap.end()) {
                    has_n |= c.n != 0; has_t |= c.t != 0;
                    it = remap.emplace(c, (uint32_t)order.size()).first;
                    order.push_back(c);
                }
                ids.push_back(it->second);
            }
            for (size_t i = 1; i + 1 < ids.size(); i++) {          // fan (obj.cpp:433-440)
                tri_idx.insert(tri_idx.end(), {ids[0], ids[i], ids[i + 1], (uint32_t)f.material});
            }
        }
        if (tri_idx.empty()) continue;
        const uint32_t vbase = (uint32_t)mesh.vertices.size();
        const size_t tbase = mesh.num_tris();
        for (size_t i = 0; i < tri_idx.size(); i += 4)
            mesh.indices.insert(mesh.indices.end(), {tri_idx[i] + vbase, tri_idx[i + 1] + vbase, tri_idx[i + 2] + vbase, tri_idx[i + 3]});
        for (auto& c : order) {
            mesh.vertices.push_back(pos[c.v]);
            mesh.texcoords.push_back(has_t ? tex[c.t] : V2());
            mesh.normals.push_back(has_n ? nrm[c.n] : V3());
        }
        for (size_t i = tbase; i < mesh.num_tris(); i++) {
            const Triangle t = mesh.tri(i);
            mesh.face_normals.push_back(normalize(cross(t.v1 - t.v0, t.v2 - t.v0)));
        }
        if (!has_n) {                                            // obj.cpp:400-411 smooth normals
            for (size_t i = tbase; i < mesh.num_tris(); i++)
                for (int k = 0; k < 3; k++) mesh.normals[mesh.indices[4 * i + k]] += mesh.face_normals[i];
        }
    }
    for (auto& n : mesh.normals) {                               // obj.cpp:492-503
        const float l2 = dot(n, n);
        if (l2 <= std::numeric_limits<float>::epsilon() || std::isnan(l2)) n = V3(0, 1, 0);
        else n = n * (1.0f / std::sqrt(l2));
    }
    return true;
}

bool load_mtl(const std::string& path, std::unordered_map<std::string, Material>& lib) {
    std::ifstream in(path);
    if (!in) { std::cerr << "Cannot open MTL file '" << path << "'" << std::endl; return false; }
    std::string line, cur;
    int errors = 0, line_no = 0;
    auto rest = [](std::istringstream& ls) { std::string s; std::getline(ls, s); auto b = s.find_first_not_of(" \t");
        auto e = s.find_last_not_of(" \t\r\n"); return b == std::string::npos ? std::string() : s.substr(b, e - b + 1); };
    while (std::getline(in, line)) {
        line_no++;
        std::istringstream ls(line);
        std::string cmd;
        if (!(ls >> cmd) || cmd[0] == '#') continue;
        if (cmd == "newmtl") {
            ls >> cur;
            if (lib.count(cur)) { std::cerr << "Material redefinition for '" << cur << "' (line " << line_no << ")." << std::endl;
                errors++; }
            lib[cur];
            continue;
        }
        Material& m = lib[cur];
        if      (cmd == "Ka") ls >> m.ka.x >> m.ka.y >> m.ka.z;
        else if (cmd == "Kd") ls >> m.kd.x >> m.kd.y >> m.kd.z;
        else if (cmd == "Ks") ls >> m.ks.x >> m.ks.y >> m.ks.z;
        else if (cmd == "Ke") ls >> m.ke.x >> m.ke.y >> m.ke.z;
        else if (cmd == "Ns") ls >> m.ns;
        else if (cmd == "Ni") ls >> m.ni;
        else if (cmd == "Tf") ls >> m.tf.x >> m.tf.y >> m.tf.z;
        else if (cmd == "Tr") ls >> m.tr;
        else if (cmd == "d")  ls >> m.d;
        else if (cmd == "illum") { float f = 0; ls >> f; m.illum = (int)f; }
        else if (cmd == "map_Ka") m.map_ka = rest(ls);
        else if (cmd == "map_Kd") m.map_kd = rest(ls);
        else if (cmd == "map_Ks") m.map_ks = rest(ls);
        else if (cmd == "map_Ke") m.map_ke = rest(ls);
        else if (cmd == "map_bump" || cmd == "bump") m.map_bump = rest(ls);
        else if (cmd == "map_d") m.map_d = rest(ls);
        else std::clog << "Unknown command '" << cmd << "' (line " << line_no << ")." << std::endl;
    }
    return errors == 0;
}

bool save_obj(const std::string& path, const TriMesh& mesh) {
    FILE* f = fopen(path.c_str(), "w");
    if (!f) return false;
    for (auto& l : mesh.mtl_libs) fprintf(f, "mtllib %s\n", l.c_str());
    for (auto& v : mesh.vertices) fprintf(f, "v %.9g %.9g %.9g\n", v.x, v.y, v.z);
    int cur = -1;
    for (size_t i = 0; i < mesh.num_tris(); i++) {
        const int m = (int)mesh.indices[4 * i + 3];
        if (m != cur && m < (int)mesh.material_names.size() && !mesh.material_names[m].empty()) {
            fprintf(f, "usemtl %s\n", mesh.material_names[m].c_str()); cur = m;
        }
        fprintf(f, "f %u %u %u\n", mesh.indices[4 * i] + 1, mesh.indices[4 * i + 1] + 1, mesh.indices[4 * i + 2] + 1);
    }
    fclose(f);
    return true;
}

} // namespace rodent
