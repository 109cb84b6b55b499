// ref_ply_shim.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" face over the REAL reference
// PLY parser so that tests can pin this repo's own Ply/GaussianCloud loader against it.
//
// This file is ours; it is compiled TOGETHER with the reference's own, unmodified sources
// where they lie (/root/reference/src/ply.cpp, core/binaryattribute.cpp, core/log.cpp) by
// oracle/Makefile into oracle/_ref/libref_ply.so.  No reference source is copied here.
// Interface used: class Ply (src/ply.h:19-46), BinaryAttribute (src/core/binaryattribute.h:12-111).
#include <cstring>
#include <fstream>
#include <string>

#include "ply.h"

extern "C" {

struct ref_ply {
    Ply ply;
    size_t vertexSize = 0;
};

// returns NULL on open/parse failure (mirrors GaussianCloud::ImportPly's use, gaussiancloud.cpp:142-159)
ref_ply* ref_ply_open(const char* path)
{
    std::ifstream f(path, std::ios::binary);
    if (!f.is_open()) return nullptr;
    ref_ply* r = new ref_ply;
    if (!r->ply.Parse(f)) {
        delete r;
        return nullptr;
    }
    r->ply.ForEachVertex([r](const void*, size_t sz) { r->vertexSize = sz; });
    return r;
}

void ref_ply_close(ref_ply* r) { delete r; }

unsigned long long ref_ply_vertex_count(const ref_ply* r) { return r->ply.GetVertexCount(); }
unsigned long long ref_ply_vertex_size(const ref_ply* r) { return r->vertexSize; }

// returns 1 and fills type (BinaryAttribute::Type as int), size, offset when the property exists
int ref_ply_get_property(const ref_ply* r, const char* name, int* type, unsigned long long* size,
                         unsigned long long* offset)
{
    BinaryAttribute a;
    if (!r->ply.GetProperty(name, a)) return 0;
    *type = (int)a.type;
    *size = a.size;
    *offset = a.offset;
    return 1;
}

// copies the raw vertex block (vertex_count * vertex_size bytes) into dst
void ref_ply_copy_vertices(const ref_ply* r, void* dst)
{
    unsigned char* out = static_cast<unsigned char*>(dst);
    r->ply.ForEachVertex([&out](const void* v, size_t sz) {
        std::memcpy(out, v, sz);
        out += sz;
    });
}

}  // extern "C"
