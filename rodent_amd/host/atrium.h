#pragma once
#include <cstdint>
#include "mesh.h"

namespace rodent {
// Fills `mesh` with the seeded procedural atrium (see atrium.cpp).  Vertices are
// not shared between parts; normals are left empty (the OBJ round trip rebuilds them).
void generate_atrium(TriMesh& mesh, uint64_t seed);
const char* atrium_mtl_text();
} // namespace rodent
