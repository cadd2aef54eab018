// npz.hip -- scipy.sparse.save_npz files read natively (the reference loads its index shards with
// scipy.sparse.load_npz + `[:, shift:]`, index.py:172-175).  Host code only: a .npz is a ZIP archive (stored or deflated members,
// ZIP64 for members beyond 4 GB) of .npy arrays `indptr`, `indices`, `data`, `shape`, `format`.  The rows are handed to
// vs_index_append_csr; nothing here touches the GPU by itself.
#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace vs {
namespace {

struct Member {
    std::string name;
    uint16_t method = 0;
    uint64_t comp_size = 0, size = 0, local_off = 0;
};

struct Npy {
    std::vector<char> raw;          // the whole decompressed .npy
    size_t data_off = 0;
    std::string descr;              // e.g. "<i4"
    std::vector<int64_t> shape;
    const char* data() const { return raw.data() + data_off; }
    int64_t count() const {                 // -1: a negative dimension or a product beyond int64 (parse_npy refuses such arrays)
        int64_t n = 1;
        for (int64_t s : shape) {
            if (s < 0 || (s != 0 && n > INT64_MAX / s)) return -1;
            n *= s;
        }
        return n;
    }
    size_t item() const { return descr.size() >= 3 ? (size_t)atoi(descr.c_str() + 2) : 0; }
};

inline uint16_t rd16(const unsigned char* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
inline uint32_t rd32(const unsigned char* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint64_t rd64(const unsigned char* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

struct File {
    FILE* f = nullptr;
    uint64_t size = 0;
    ~File() { if (f) fclose(f); }
    int open(const char* path) {
        f = fopen(path, "rb");
        if (!f) return fail(VS_EINVAL, "cannot open %s", path);
        if (fseeko(f, 0, SEEK_END) != 0) return fail(VS_EINVAL, "%s: cannot seek", path);
        size = (uint64_t)ftello(f);
        return VS_OK;
    }
    int read(uint64_t off, void* dst, size_t n) {
        if (off + n > size) return fail(VS_EINVAL, "zip: read beyond the end of the file");
        if (fseeko(f, (off_t)off, SEEK_SET) != 0 || fread(dst, 1, n, f) != n) return fail(VS_EINVAL, "zip: read failed");
        return VS_OK;
    }
};

// central directory -> members (ZIP64 aware)
int list_members(File& zf, std::vector<Member>& out) {
    const uint64_t tail = std::min<uint64_t>(zf.size, 65536 + 22 + 20);
    std::vector<unsigned char> buf(tail);
    VS_TRY(zf.read(zf.size - tail, buf.data(), tail));
    int64_t eocd = -1;
    for (int64_t i = (int64_t)tail - 22; i >= 0; --i)
        if (rd32(&buf[i]) == 0x06054b50u) { eocd = i; break; }
    if (eocd < 0) return fail(VS_EINVAL, "not a zip archive (no end-of-central-directory record)");
    uint64_t n_entries = rd16(&buf[eocd + 10]), cd_size = rd32(&buf[eocd + 12]), cd_off = rd32(&buf[eocd + 16]);
    if (n_entries == 0xFFFF || cd_size == 0xFFFFFFFFu || cd_off == 0xFFFFFFFFu) {
        if (eocd < 20 || rd32(&buf[eocd - 20]) != 0x07064b50u) return fail(VS_EINVAL, "zip64 locator missing");
        const uint64_t e64 = rd64(&buf[eocd - 20 + 8]);
        unsigned char r[56];
        VS_TRY(zf.read(e64, r, 56));
        if (rd32(r) != 0x06064b50u) return fail(VS_EINVAL, "zip64 end-of-central-directory record missing");
        n_entries = rd64(r + 32);
        cd_size = rd64(r + 40);
        cd_off = rd64(r + 48);
    }
    if (cd_size > zf.size || cd_off > zf.size - cd_size || n_entries > cd_size / 46 + 1) return fail(VS_EINVAL, "zip: central directory beyond the end of the file");
    std::vector<unsigned char> cd(cd_size);
    VS_TRY(zf.read(cd_off, cd.data(), cd_size));
    size_t p = 0;
    for (uint64_t e = 0; e < n_entries; ++e) {
        if (p + 46 > cd.size() || rd32(&cd[p]) != 0x02014b50u) return fail(VS_EINVAL, "zip: bad central directory entry");
        Member m;
        m.method = rd16(&cd[p + 10]);
        m.comp_size = rd32(&cd[p + 20]);
        m.size = rd32(&cd[p + 24]);
        const uint16_t nlen = rd16(&cd[p + 28]), xlen = rd16(&cd[p + 30]), clen = rd16(&cd[p + 32]);
        m.local_off = rd32(&cd[p + 42]);
        if (p + 46 + nlen + xlen + clen > cd.size()) return fail(VS_EINVAL, "zip: truncated central directory");
        m.name.assign(reinterpret_cast<const char*>(&cd[p + 46]), nlen);
        size_t x = p + 46 + nlen;
        const size_t xend = x + xlen;
        while (x + 4 <= xend) {                                  // zip64 extended information: the fields that overflowed, in order
            const uint16_t id = rd16(&cd[x]), sz = rd16(&cd[x + 2]);
            if (id == 0x0001) {
                size_t q = x + 4;
                if (m.size == 0xFFFFFFFFu && q + 8 <= xend) { m.size = rd64(&cd[q]); q += 8; }
                if (m.comp_size == 0xFFFFFFFFu && q + 8 <= xend) { m.comp_size = rd64(&cd[q]); q += 8; }
                if (m.local_off == 0xFFFFFFFFu && q + 8 <= xend) { m.local_off = rd64(&cd[q]); q += 8; }
            }
            x += 4 + (size_t)sz;
        }
        out.push_back(m);
        p += 46 + (size_t)nlen + xlen + clen;
    }
    return VS_OK;
}

int read_member(File& zf, const Member& m, std::vector<char>& out) {
    unsigned char lh[30];
    VS_TRY(zf.read(m.local_off, lh, 30));
    if (rd32(lh) != 0x04034b50u) return fail(VS_EINVAL, "zip: bad local header of %s", m.name.c_str());
    const uint64_t data_off = m.local_off + 30 + rd16(lh + 26) + rd16(lh + 28);
    // untrusted sizes: the compressed bytes must lie inside the file, and deflate expands at most ~1032 x
    if (data_off > zf.size || m.comp_size > zf.size - data_off) return fail(VS_EINVAL, "zip: member %s lies beyond the end of the file", m.name.c_str());
    if (m.size > (uint64_t)1 << 46 || (m.method == 8 && m.size / 1040 > m.comp_size + 1)) return fail(VS_EINVAL, "zip: member %s claims an impossible size", m.name.c_str());
    out.resize(m.size);
    if (m.method == 0) {
        if (m.comp_size != m.size) return fail(VS_EINVAL, "zip: stored member %s with differing sizes", m.name.c_str());
        return zf.read(data_off, out.data(), m.size);
    }
    if (m.method != 8) return fail(VS_EUNSUPPORTED, "zip: member %s uses compression method %d", m.name.c_str(), (int)m.method);
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, -15) != Z_OK) return fail(VS_EINVAL, "zlib: inflateInit2 failed");       // raw deflate stream
    out.resize(m.size + 8);                                                                          // (+ 8: a stream longer than its header shows up as overflow)
    std::vector<unsigned char> in((size_t)std::max<uint64_t>(1, std::min<uint64_t>(m.comp_size, (uint64_t)8 << 20)));
    uint64_t consumed = 0, produced = 0;
    for (;;) {
        if (zs.avail_in == 0 && consumed < m.comp_size) {
            const size_t n = (size_t)std::min<uint64_t>(in.size(), m.comp_size - consumed);
            if (zf.read(data_off + consumed, in.data(), n) != VS_OK) { inflateEnd(&zs); return VS_EINVAL; }
            consumed += n;
            zs.next_in = in.data();
            zs.avail_in = (uInt)n;
        }
        if (zs.avail_out == 0) {
            const uint64_t room = (uint64_t)out.size() - produced;
            if (room == 0) { inflateEnd(&zs); return fail(VS_EINVAL, "zip: %s is longer than its header says", m.name.c_str()); }
            zs.next_out = reinterpret_cast<unsigned char*>(out.data()) + produced;
            zs.avail_out = (uInt)std::min<uint64_t>(room, (uint64_t)1 << 30);
        }
        const uInt before = zs.avail_out;
        const int zr = inflate(&zs, Z_NO_FLUSH);
        produced += before - zs.avail_out;
        if (zr == Z_STREAM_END) break;
        if (zr != Z_OK && !(zr == Z_BUF_ERROR && (zs.avail_out == 0 || (zs.avail_in == 0 && consumed < m.comp_size)))) {
            inflateEnd(&zs);
            return fail(VS_EINVAL, "zlib: inflate of %s failed (%d)%s", m.name.c_str(), zr, consumed >= m.comp_size ? ": truncated member" : "");
        }
    }
    inflateEnd(&zs);
    if (produced != m.size) return fail(VS_EINVAL, "zip: %s inflates to %llu bytes, the directory says %llu", m.name.c_str(), (unsigned long long)produced,
                                        (unsigned long long)m.size);
    out.resize(m.size);
    return VS_OK;
}

// .npy header: magic, version, little-endian dict {'descr': '<i4', 'fortran_order': False, 'shape': (N,), }
int parse_npy(Npy& a, const char* what) {
    const std::vector<char>& r = a.raw;
    if (r.size() < 10 || memcmp(r.data(), "\x93NUMPY", 6) != 0) return fail(VS_EINVAL, "%s: not a .npy array", what);
    const int major = (unsigned char)r[6];
    size_t hlen, hoff;
    if (major == 1) { hlen = rd16(reinterpret_cast<const unsigned char*>(&r[8])); hoff = 10; }
    else if (major == 2 || major == 3) { if (r.size() < 12) return fail(VS_EINVAL, "%s: short .npy", what); hlen = rd32(reinterpret_cast<const unsigned char*>(&r[8])); hoff = 12; }
    else return fail(VS_EUNSUPPORTED, "%s: .npy format version %d", what, major);
    if (hoff + hlen > r.size()) return fail(VS_EINVAL, "%s: truncated .npy header", what);
    const std::string h(r.data() + hoff, hlen);
    auto find_val = [&](const char* key) -> size_t {
        const size_t k = h.find(key);
        if (k == std::string::npos) return k;
        const size_t c = h.find(':', k);
        return c == std::string::npos ? c : c + 1;
    };
    size_t p = find_val("'descr'");
    if (p == std::string::npos) return fail(VS_EINVAL, "%s: .npy header without descr", what);
    const size_t q0 = h.find('\'', p), q1 = h.find('\'', q0 + 1);
    if (q0 == std::string::npos || q1 == std::string::npos) return fail(VS_EUNSUPPORTED, "%s: structured .npy dtypes are not supported", what);
    a.descr = h.substr(q0 + 1, q1 - q0 - 1);
    p = find_val("'fortran_order'");
    if (p != std::string::npos) {
        const size_t v = h.find_first_not_of(' ', p);
        if (v != std::string::npos && h.compare(v, 4, "True") == 0) return fail(VS_EUNSUPPORTED, "%s: Fortran-ordered .npy", what);
    }
    p = find_val("'shape'");
    if (p == std::string::npos) return fail(VS_EINVAL, "%s: .npy header without shape", what);
    const size_t s0 = h.find('(', p), s1 = h.find(')', s0);
    if (s0 == std::string::npos || s1 == std::string::npos) return fail(VS_EINVAL, "%s: bad shape", what);
    a.shape.clear();
    const std::string dims = h.substr(s0 + 1, s1 - s0 - 1);
    size_t i = 0;
    while (i < dims.size()) {
        while (i < dims.size() && (dims[i] == ' ' || dims[i] == ',')) ++i;
        if (i >= dims.size()) break;
        a.shape.push_back(strtoll(dims.c_str() + i, nullptr, 10));
        while (i < dims.size() && dims[i] != ',') ++i;
    }
    a.data_off = hoff + hlen;
    if (a.descr.size() < 3 || (a.descr[0] != '<' && a.descr[0] != '|' && a.descr[0] != '=')) return fail(VS_EUNSUPPORTED, "%s: dtype %s (big-endian?)", what, a.descr.c_str());
    // untrusted file: the item size must be one the readers below handle, the shape must not overflow, the payload must be there
    const size_t isz = a.item();
    const char kind = a.descr[1];
    const bool known = ((kind == 'i' || kind == 'u') && (isz == 1 || isz == 2 || isz == 4 || isz == 8)) || (kind == 'b' && isz == 1) ||
                       (kind == 'f' && (isz == 2 || isz == 4 || isz == 8)) || ((kind == 'U' || kind == 'S') && isz >= 1 && isz <= 64);
    if (!known) return fail(VS_EUNSUPPORTED, "%s: dtype %s is not supported", what, a.descr.c_str());
    if (a.shape.size() > 2) return fail(VS_EINVAL, "%s: %zu-dimensional array", what, a.shape.size());
    const int64_t cnt = a.count();
    const size_t esz = (kind == 'U') ? isz * 4 : isz;              // ('<U3' = 3 UCS-4 characters)
    if (cnt < 0 || a.data_off > r.size() || (uint64_t)cnt > (r.size() - a.data_off) / esz) return fail(VS_EINVAL, "%s: .npy payload shorter than its shape", what);
    return VS_OK;
}

int64_t int_at(const Npy& a, int64_t i) {
    const char* p = a.data() + (size_t)i * a.item();
    const char k = a.descr[1];
    switch (a.item()) {
        case 1: return k == 'u' || k == 'b' ? (int64_t)*reinterpret_cast<const uint8_t*>(p) : (int64_t)*reinterpret_cast<const int8_t*>(p);
        case 2: return k == 'u' ? (int64_t)*reinterpret_cast<const uint16_t*>(p) : (int64_t)*reinterpret_cast<const int16_t*>(p);
        case 4: return k == 'u' ? (int64_t)*reinterpret_cast<const uint32_t*>(p) : (int64_t)*reinterpret_cast<const int32_t*>(p);
        default: return *reinterpret_cast<const int64_t*>(p);
    }
}

float half_to_float(uint16_t h) {
    const uint32_t s = (uint32_t)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    uint32_t u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int sh = 0; uint32_t mm = m; while (!(mm & 1024u)) { mm <<= 1; ++sh; } u = s | ((uint32_t)(113 - sh) << 23) | ((mm & 1023u) << 13); }
    } else if (e == 31) u = s | 0x7F800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f;
    memcpy(&f, &u, 4);
    return f;
}

float value_at(const Npy& a, int64_t i) {
    const char* p = a.data() + (size_t)i * a.item();
    const char k = a.descr[1];
    if (k == 'f') {
        if (a.item() == 4) return *reinterpret_cast<const float*>(p);
        if (a.item() == 8) return (float)*reinterpret_cast<const double*>(p);
        if (a.item() == 2) return half_to_float(*reinterpret_cast<const uint16_t*>(p));
    }
    return (float)int_at(a, i);
}

struct CsrFile {
    int64_t n_rows = 0, n_cols = 0;
    Npy indptr, indices, data;
    bool has_data = false;
};

int load_csr_members(const char* path, bool want_payload, CsrFile& c) {
    File zf;
    VS_TRY(zf.open(path));
    std::vector<Member> ms;
    VS_TRY(list_members(zf, ms));
    auto find = [&](const char* stem) -> const Member* {
        const std::string n = std::string(stem) + ".npy";
        for (const Member& m : ms) if (m.name == n) return &m;
        return nullptr;
    };
    const Member* m_fmt = find("format");
    if (m_fmt) {
        Npy f;
        VS_TRY(read_member(zf, *m_fmt, f.raw));
        VS_TRY(parse_npy(f, "format"));
        const size_t fesz = f.descr[1] == 'U' ? f.item() * 4 : f.item();
        const std::string v(f.data(), std::min<size_t>(fesz * (size_t)std::max<int64_t>(1, f.count()), 16));
        // '|S3' b"csr" or '<U3' "csr" (4 bytes per character)
        std::string flat;
        for (char ch : v) if (ch) flat.push_back(ch);
        if (flat != "csr") return fail(VS_EUNSUPPORTED, "%s holds a '%s' matrix: only CSR files are read natively", path, flat.c_str());
    }
    const Member* m_shape = find("shape");
    const Member* m_ptr = find("indptr");
    const Member* m_idx = find("indices");
    const Member* m_dat = find("data");
    if (!m_shape || !m_ptr || !m_idx) return fail(VS_EINVAL, "%s: not a scipy.sparse .npz (shape / indptr / indices missing)", path);
    Npy sh;
    VS_TRY(read_member(zf, *m_shape, sh.raw));
    VS_TRY(parse_npy(sh, "shape"));
    if (sh.count() != 2 || (sh.descr[1] != 'i' && sh.descr[1] != 'u')) return fail(VS_EINVAL, "%s: shape is not a pair of integers", path);
    c.n_rows = int_at(sh, 0);
    c.n_cols = int_at(sh, 1);
    VS_TRY(read_member(zf, *m_ptr, c.indptr.raw));
    VS_TRY(parse_npy(c.indptr, "indptr"));
    if (c.indptr.count() != c.n_rows + 1) return fail(VS_EINVAL, "%s: indptr has %lld entries for %lld rows", path, (long long)c.indptr.count(), (long long)c.n_rows);
    if (c.indptr.descr[1] != 'i' && c.indptr.descr[1] != 'u') return fail(VS_EINVAL, "%s: indptr is not an integer array", path);
    VS_TRY(read_member(zf, *m_idx, c.indices.raw));
    VS_TRY(parse_npy(c.indices, "indices"));
    if (c.indices.descr[1] != 'i' && c.indices.descr[1] != 'u') return fail(VS_EINVAL, "%s: indices is not an integer array", path);
    if (c.n_rows < 0 || c.n_cols < 0 || c.indptr.shape.size() != 1 || c.indices.shape.size() != 1) return fail(VS_EINVAL, "%s: indptr / indices must be 1-d", path);
    const int64_t nnz = int_at(c.indptr, c.n_rows);
    if (int_at(c.indptr, 0) != 0 || nnz != c.indices.count()) return fail(VS_EINVAL, "%s: indptr does not match indices (%lld vs %lld)", path, (long long)nnz, (long long)c.indices.count());
    // the WHOLE row-pointer array before anybody indexes `indices` with it: monotone, inside [0, nnz]
    for (int64_t r = 0; r < c.n_rows; ++r) {
        const int64_t a = int_at(c.indptr, r), b = int_at(c.indptr, r + 1);
        if (a < 0 || b < a || b > nnz) return fail(VS_EINVAL, "%s: indptr is not a row-pointer array at row %lld (%lld, %lld; %lld non-zeros)", path, (long long)r, (long long)a, (long long)b, (long long)nnz);
    }
    if (m_dat && want_payload) {
        VS_TRY(read_member(zf, *m_dat, c.data.raw));
        VS_TRY(parse_npy(c.data, "data"));
        if (c.data.shape.size() != 1 || c.data.descr[1] == 'U' || c.data.descr[1] == 'S') return fail(VS_EINVAL, "%s: data must be a 1-d numeric array", path);
        if (c.data.count() != nnz) return fail(VS_EINVAL, "%s: data has %lld entries for %lld non-zeros", path, (long long)c.data.count(), (long long)nnz);
        c.has_data = true;
    }
    return VS_OK;
}

}  // namespace
}  // namespace vs

using namespace vs;

// scipy.sparse.save_npz file (CSR): shape, non-zeros and 8-nnz packets that remain after `[:, shift:]` (index.py:172)
static int npz_inspect_impl(const char* path, int32_t shift, int64_t* n_rows, int64_t* n_cols, int64_t* nnz, int64_t* packets) {
    if (!path || shift < 0) return fail(VS_EINVAL, "bad argument");
    CsrFile c;
    VS_TRY(load_csr_members(path, false, c));
    if (shift > c.n_cols) return fail(VS_EINVAL, "shift %d beyond the %lld columns of %s", shift, (long long)c.n_cols, path);
    int64_t kept = 0, pk = 0;
    for (int64_t r = 0; r < c.n_rows; ++r) {
        const int64_t a = int_at(c.indptr, r), b = int_at(c.indptr, r + 1);
        if (b < a) return fail(VS_EINVAL, "%s: indptr decreases at row %lld", path, (long long)r);
        int64_t len = 0;
        for (int64_t i = a; i < b; ++i) {
            const int64_t col = int_at(c.indices, i);
            if (col < 0 || col >= c.n_cols) return fail(VS_EINVAL, "%s: column id %lld out of range in row %lld", path, (long long)col, (long long)r);
            len += col >= shift ? 1 : 0;
        }
        kept += len;
        pk += (len + 7) / 8;
    }
    if (n_rows) *n_rows = c.n_rows;
    if (n_cols) *n_cols = c.n_cols - shift;
    if (nnz) *nnz = kept;
    if (packets) *packets = pk;
    return VS_OK;
}

// appends the rows of a scipy .npz CSR shard to a reserved index: columns below `shift` dropped, ids moved down by `shift`,
// columns sorted within a row (what `load_npz(f)[:, shift:]` + sort_indices() gives the reference's vstack, index.py:172-175)
static int index_append_npz_impl(vs_index* idx, const char* path, int32_t shift) {
    if (!idx || !path || shift < 0) return fail(VS_EINVAL, "bad argument");
    if (idx->kind != VS_KIND_CSR) return fail(VS_EINVAL, "not a CSR index");
    CsrFile c;
    VS_TRY(load_csr_members(path, true, c));
    if (c.n_cols - shift != idx->n_cols) return fail(VS_EINVAL, "%s has %lld columns after the shift, the index %d", path, (long long)(c.n_cols - shift), idx->n_cols);
    const bool binary = idx->store_dtype == VS_NONE;
    std::vector<int64_t> rp((size_t)c.n_rows + 1);
    std::vector<int32_t> cols;
    std::vector<float> vals;
    cols.reserve((size_t)c.indices.count());
    if (!binary) vals.reserve((size_t)c.indices.count());
    std::vector<std::pair<int32_t, float>> row;
    rp[0] = 0;
    for (int64_t r = 0; r < c.n_rows; ++r) {
        const int64_t a = int_at(c.indptr, r), b = int_at(c.indptr, r + 1);
        if (b < a) return fail(VS_EINVAL, "%s: indptr decreases at row %lld", path, (long long)r);
        row.clear();
        bool sorted = true;
        for (int64_t i = a; i < b; ++i) {
            const int64_t col = int_at(c.indices, i);
            if (col < 0 || col >= c.n_cols) return fail(VS_EINVAL, "%s: column id %lld out of range in row %lld", path, (long long)col, (long long)r);
            if (col < shift) continue;
            const float v = c.has_data ? value_at(c.data, i) : 1.f;
            if (binary && v != 1.f) return fail(VS_EINVAL, "BoTIndex expects a binary matrix (every stored value == 1): %s row %lld holds %g", path, (long long)r, (double)v);
            if (!row.empty() && (int32_t)(col - shift) < row.back().first) sorted = false;
            row.emplace_back((int32_t)(col - shift), v);
        }
        if (!sorted) std::stable_sort(row.begin(), row.end(), [](const auto& x, const auto& y) { return x.first < y.first; });
        for (const auto& e : row) {
            cols.push_back(e.first);
            if (!binary) vals.push_back(e.second);
        }
        rp[(size_t)r + 1] = (int64_t)cols.size();
    }
    return vs_index_append_csr(idx, rp.data(), VS_I64, cols.data(), VS_I32, binary ? nullptr : vals.data(), VS_F32, c.n_rows);
}

// ---- writer: SparseIndex.save (index.py:181-202: CSR -> scipy save_npz) without scipy --------------------------------------------
namespace vs {
namespace {

struct ZipWriter {
    FILE* f = nullptr;
    struct Entry { std::string name; uint32_t crc; uint64_t comp, size, off; uint16_t method; };
    std::vector<Entry> entries;
    uint64_t pos = 0;
    ~ZipWriter() { if (f) fclose(f); }
    int put(const void* p, size_t n) {
        if (n && fwrite(p, 1, n, f) != n) return fail(VS_EINVAL, "zip: write failed (disk full?)");
        pos += n;
        return VS_OK;
    }
    static void w16(std::vector<unsigned char>& b, uint16_t v) { b.push_back((unsigned char)v); b.push_back((unsigned char)(v >> 8)); }
    static void w32(std::vector<unsigned char>& b, uint32_t v) { for (int i = 0; i < 4; ++i) b.push_back((unsigned char)(v >> (8 * i))); }
    static void w64(std::vector<unsigned char>& b, uint64_t v) { for (int i = 0; i < 8; ++i) b.push_back((unsigned char)(v >> (8 * i))); }

    // one member = .npy header + payload; sizes are known up front (stored) or patched through a zip64 data descriptor-free layout:
    // deflated members are compressed into memory chunk by chunk, so the local header can carry the final sizes
    int add(const std::string& name, const std::vector<char>& head, const void* payload, uint64_t payload_bytes, bool deflate_it) {
        const uint64_t size = head.size() + payload_bytes;
        uint32_t crc = (uint32_t)crc32(0L, Z_NULL, 0);
        auto crc_feed = [&](const void* p, uint64_t n) {
            const unsigned char* q = reinterpret_cast<const unsigned char*>(p);
            while (n) { const uInt c = (uInt)std::min<uint64_t>(n, (uint64_t)1 << 30); crc = (uint32_t)crc32(crc, q, c); q += c; n -= c; }
        };
        crc_feed(head.data(), head.size());
        crc_feed(payload, payload_bytes);
        std::vector<unsigned char> comp;
        if (deflate_it) {
            z_stream zs;
            memset(&zs, 0, sizeof(zs));
            if (deflateInit2(&zs, 1, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return fail(VS_EINVAL, "zlib: deflateInit2 failed");
            comp.resize((size_t)(size + size / 1000 + 1024));
            uint64_t produced = 0;
            auto feed = [&](const void* p, uint64_t n, int last) -> int {
                const unsigned char* q = reinterpret_cast<const unsigned char*>(p);
                do {
                    const uInt c = (uInt)std::min<uint64_t>(n, (uint64_t)1 << 30);
                    zs.next_in = const_cast<unsigned char*>(q);
                    zs.avail_in = c;
                    q += c; n -= c;
                    const int flush = (last && n == 0) ? Z_FINISH : Z_NO_FLUSH;
                    for (;;) {
                        if (produced == comp.size()) comp.resize(comp.size() + comp.size() / 4 + 65536);
                        const uInt room = (uInt)std::min<uint64_t>(comp.size() - produced, (uint64_t)1 << 30);
                        zs.next_out = comp.data() + produced;
                        zs.avail_out = room;
                        const int zr = deflate(&zs, flush);
                        produced += room - zs.avail_out;
                        if (zr == Z_STREAM_END) break;
                        if (zr != Z_OK && zr != Z_BUF_ERROR) return fail(VS_EINVAL, "zlib: deflate failed (%d)", zr);
                        if (zs.avail_in == 0 && zs.avail_out != 0 && flush != Z_FINISH) break;
                    }
                } while (n);
                return VS_OK;
            };
            int rc = feed(head.data(), head.size(), payload_bytes == 0);
            if (rc == VS_OK && payload_bytes) rc = feed(payload, payload_bytes, 1);
            deflateEnd(&zs);
            if (rc != VS_OK) return rc;
            comp.resize((size_t)produced);
        }
        Entry e{name, crc, deflate_it ? (uint64_t)comp.size() : size, size, pos, (uint16_t)(deflate_it ? 8 : 0)};
        const bool z64 = e.size >= 0xFFFFFFFFull || e.comp >= 0xFFFFFFFFull;
        std::vector<unsigned char> lh;
        w32(lh, 0x04034b50u); w16(lh, z64 ? 45 : 20); w16(lh, 0); w16(lh, e.method); w16(lh, 0); w16(lh, 0x21);   // (fixed DOS date 1980-01-01)
        w32(lh, e.crc); w32(lh, z64 ? 0xFFFFFFFFu : (uint32_t)e.comp); w32(lh, z64 ? 0xFFFFFFFFu : (uint32_t)e.size);
        w16(lh, (uint16_t)name.size()); w16(lh, z64 ? 20 : 0);
        lh.insert(lh.end(), name.begin(), name.end());
        if (z64) { w16(lh, 0x0001); w16(lh, 16); w64(lh, e.size); w64(lh, e.comp); }
        VS_TRY(put(lh.data(), lh.size()));
        if (deflate_it) VS_TRY(put(comp.data(), comp.size()));
        else { VS_TRY(put(head.data(), head.size())); VS_TRY(put(payload, (size_t)payload_bytes)); }
        entries.push_back(e);
        return VS_OK;
    }
    int finish() {
        const uint64_t cd_off = pos;
        std::vector<unsigned char> cd;
        for (const Entry& e : entries) {
            const bool z64 = e.size >= 0xFFFFFFFFull || e.comp >= 0xFFFFFFFFull || e.off >= 0xFFFFFFFFull;
            w32(cd, 0x02014b50u); w16(cd, 45); w16(cd, z64 ? 45 : 20); w16(cd, 0); w16(cd, e.method); w16(cd, 0); w16(cd, 0x21);
            w32(cd, e.crc); w32(cd, z64 ? 0xFFFFFFFFu : (uint32_t)e.comp); w32(cd, z64 ? 0xFFFFFFFFu : (uint32_t)e.size);
            w16(cd, (uint16_t)e.name.size()); w16(cd, z64 ? 28 : 0); w16(cd, 0); w16(cd, 0); w16(cd, 0); w32(cd, 0);
            w32(cd, z64 ? 0xFFFFFFFFu : (uint32_t)e.off);
            cd.insert(cd.end(), e.name.begin(), e.name.end());
            if (z64) { w16(cd, 0x0001); w16(cd, 24); w64(cd, e.size); w64(cd, e.comp); w64(cd, e.off); }
        }
        VS_TRY(put(cd.data(), cd.size()));
        const uint64_t cd_size = cd.size();
        std::vector<unsigned char> tail;
        if (cd_off >= 0xFFFFFFFFull || cd_size >= 0xFFFFFFFFull) {
            const uint64_t e64 = pos;
            w32(tail, 0x06064b50u); w64(tail, 44); w16(tail, 45); w16(tail, 45); w32(tail, 0); w32(tail, 0);
            w64(tail, entries.size()); w64(tail, entries.size()); w64(tail, cd_size); w64(tail, cd_off);
            w32(tail, 0x07064b50u); w32(tail, 0); w64(tail, e64); w32(tail, 1);
        }
        w32(tail, 0x06054b50u); w16(tail, 0); w16(tail, 0); w16(tail, (uint16_t)entries.size()); w16(tail, (uint16_t)entries.size());
        w32(tail, cd_size >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cd_size); w32(tail, cd_off >= 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)cd_off);
        w16(tail, 0);
        VS_TRY(put(tail.data(), tail.size()));
        if (fflush(f) != 0) return fail(VS_EINVAL, "zip: flush failed");
        return VS_OK;
    }
};

std::vector<char> npy_header(const char* descr, const std::vector<int64_t>& shape) {
    std::string d = std::string("{'descr': '") + descr + "', 'fortran_order': False, 'shape': (";
    for (size_t i = 0; i < shape.size(); ++i) d += std::to_string(shape[i]) + (shape.size() == 1 ? "," : (i + 1 < shape.size() ? ", " : ""));
    d += "), }";
    size_t total = 10 + d.size() + 1;
    const size_t pad = (64 - total % 64) % 64;
    d.append(pad, ' ');
    d.push_back('\n');
    std::vector<char> h;
    const char magic[8] = {'\x93', 'N', 'U', 'M', 'P', 'Y', 1, 0};
    h.insert(h.end(), magic, magic + 8);
    h.push_back((char)(d.size() & 0xFF));
    h.push_back((char)(d.size() >> 8));
    h.insert(h.end(), d.begin(), d.end());
    return h;
}

}  // namespace
}  // namespace vs

// the index as a scipy.sparse.save_npz file (CSR; int64 indptr / indices like the reference's torch CSR, fp32 data; a binary
// index writes data == 1): scipy.sparse.load_npz reads it back.  compressed = 0: stored members (fast), 1: deflate level 1.
static int index_save_npz_impl(const vs_index* idx, const char* path, int compressed) {
    if (!idx || !path) return fail(VS_EINVAL, "NULL argument");
    if (idx->kind != VS_KIND_CSR) return fail(VS_EINVAL, "not a CSR index");
    std::vector<int64_t> rp((size_t)idx->n_rows + 1);
    VS_TRY(vs_index_export_csr(idx, rp.data(), nullptr, nullptr, VS_F32));
    const int64_t nnz = rp[(size_t)idx->n_rows];
    std::vector<int64_t> ci((size_t)std::max<int64_t>(nnz, 1));
    std::vector<float> va((size_t)std::max<int64_t>(nnz, 1), 1.f);
    VS_TRY(vs_index_export_csr(idx, rp.data(), ci.data(), idx->store_dtype == VS_NONE ? nullptr : (void*)va.data(), VS_F32));
    ZipWriter zw;
    zw.f = fopen(path, "wb");
    if (!zw.f) return fail(VS_EINVAL, "cannot create %s", path);
    const bool z = compressed != 0;
    const int64_t shape[2] = {idx->n_rows, idx->n_cols};
    VS_TRY(zw.add("indices.npy", npy_header("<i8", {nnz}), ci.data(), (uint64_t)nnz * 8, z));
    VS_TRY(zw.add("indptr.npy", npy_header("<i8", {idx->n_rows + 1}), rp.data(), (uint64_t)(idx->n_rows + 1) * 8, z));
    VS_TRY(zw.add("format.npy", npy_header("|S3", {}), "csr", 3, z));
    VS_TRY(zw.add("shape.npy", npy_header("<i8", {2}), shape, 16, z));
    VS_TRY(zw.add("data.npy", npy_header("<f4", {nnz}), va.data(), (uint64_t)nnz * 4, z));
    const unsigned char yes = 1;                                  // scipy >= 1.11 marks sparse ARRAYS (the facade holds a csr_array, like the
    VS_TRY(zw.add("_is_array.npy", npy_header("|b1", {}), &yes, 1, z));      // reference's save path under the pinned scipy: tests/golden/save_load.npz)
    return zw.finish();
}

// The C entry points: a corrupt or hostile file must come back as an error code, never as a C++ exception through the C ABI
// (std::bad_alloc / length_error from a size field, out_of_range from a header parser) -- that would terminate the caller's process.
template <class F>
static int npz_guard(F&& f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(VS_ENOMEM, "npz: out of host memory (a size field of the file?)");
    } catch (const std::exception& e) {
        return fail(VS_EINVAL, "npz: malformed file (%s)", e.what());
    } catch (...) {
        return fail(VS_EINVAL, "npz: malformed file");
    }
}
extern "C" int vs_npz_inspect(const char* path, int32_t shift, int64_t* n_rows, int64_t* n_cols, int64_t* nnz, int64_t* packets) {
    return npz_guard([&] { return npz_inspect_impl(path, shift, n_rows, n_cols, nnz, packets); });
}
extern "C" int vs_index_append_npz(vs_index* idx, const char* path, int32_t shift) {
    return npz_guard([&] { return index_append_npz_impl(idx, path, shift); });
}
extern "C" int vs_index_save_npz(const vs_index* idx, const char* path, int compressed) {
    return npz_guard([&] { return index_save_npz_impl(idx, path, compressed); });
}
