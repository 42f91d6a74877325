#pragma once
#include <cstdint>
#include "mesh.h"

namespace rodent {
// Fills `mesh` with the seeded procedural atrium (see atrium.cpp).  Vertices are
// not shared between parts; normals are left empty (the OBJ round trip rebuilds them).
// detail > 1: every curved part tessellated detail x detail times finer, detail^2 times the foliage ("gallery": detail 4, 4.2 M triangles).
void generate_atrium(TriMesh& mesh, uint64_t seed, int detail = 1);
const char* atrium_mtl_text();
// The other scene classes of the reference's benchmark suite (benchmarks/benchmark.py:16-21: sponza, crown, san-miguel, powerplant), as
// seeded stand-ins (stress_scenes.cpp): a dense organic single mesh of small triangles, and a hall of long thin triangles.
void generate_crown(TriMesh& mesh, uint64_t seed, int detail = 4);
void generate_plant(TriMesh& mesh, uint64_t seed, int detail = 4);
} // namespace rodent
