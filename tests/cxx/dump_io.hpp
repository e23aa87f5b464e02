// TEST INFRASTRUCTURE: the record files the Python tests and the C++ ABI drivers exchange.
#ifndef DUMP_IO_HPP
#define DUMP_IO_HPP
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>
// ---- dump files: records of (name, dtype 0 f64 / 1 i32 / 2 u8, count, raw bytes)
struct Blob { int dtype; std::vector<char> raw; size_t count; };
typedef std::map<std::string, Blob> Dump;
static inline bool read_dump(const char *path, Dump &d) {
    FILE *f = fopen(path, "rb"); if (!f) return false;
    for (;;) { uint32_t nl; if (fread(&nl, 4, 1, f) != 1) break;
        std::string name(nl, 0); uint8_t dt; uint64_t cnt;
        if (fread(&name[0], 1, nl, f) != nl || fread(&dt, 1, 1, f) != 1 || fread(&cnt, 8, 1, f) != 1) { fclose(f); return false; }
        Blob b; b.dtype = dt; b.count = (size_t)cnt; b.raw.resize(cnt*(dt == 0 ? 8 : dt == 1 ? 4 : 1));
        if (!b.raw.empty() && fread(b.raw.data(), 1, b.raw.size(), f) != b.raw.size()) { fclose(f); return false; }
        d[name] = b; }
    fclose(f); return true;
}
static inline void put(FILE *f, const char *name, int dt, const void *p, size_t cnt) {
    uint32_t nl = (uint32_t)strlen(name); uint8_t d = (uint8_t)dt; uint64_t c = cnt;
    fwrite(&nl, 4, 1, f); fwrite(name, 1, nl, f); fwrite(&d, 1, 1, f); fwrite(&c, 8, 1, f); fwrite(p, dt == 0 ? 8 : dt == 1 ? 4 : 1, cnt, f);
}
static inline const double *F64(const Dump &d, const std::string &n) { Dump::const_iterator it = d.find(n); return it == d.end() || it->second.raw.empty() ? nullptr : (const double *)it->second.raw.data(); }
static inline const int32_t *I32(const Dump &d, const std::string &n) { Dump::const_iterator it = d.find(n); return it == d.end() || it->second.raw.empty() ? nullptr : (const int32_t *)it->second.raw.data(); }
static inline const uint8_t *U8(const Dump &d, const std::string &n) { Dump::const_iterator it = d.find(n); return it == d.end() || it->second.raw.empty() ? nullptr : (const uint8_t *)it->second.raw.data(); }
static inline size_t CNT(const Dump &d, const std::string &n) { Dump::const_iterator it = d.find(n); return it == d.end() ? 0 : it->second.count; }
#endif
