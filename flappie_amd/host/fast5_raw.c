/*  fast5_raw.c -- the read side of a single-read fast5 file WITHOUT libhdf5 (round 6).
 *
 *  What read_raw (/root/reference/src/fast5_interface.c:231-318) takes from a file is four things: the first entry of /Raw/Reads in
 *  name order, its `read_id` attribute, its `Signal` dataset, and three numeric attributes of /UniqueGlobalKey/channel_id.  Through
 *  libhdf5 that costs ~105 us a file (profiles/r05_host_scaling.txt: H5Fopen alone 40 us -- property lists, a metadata cache, a
 *  skip list of open objects per file), and at H = 256 eight ranks of a node ask for 3.3 CPUs each of exactly that.  A single-read
 *  file is 15 ... 400 KB: this file reads it with ONE read(2) and walks the HDF5 structures in memory -- superblock (versions 0-3),
 *  object headers (versions 1 and 2, continuation blocks), old-style groups (symbol table message -> v1 B-tree -> symbol nodes ->
 *  local heap) and compact new-style groups (link messages), attribute messages (versions 1-3; fixed strings, variable-length
 *  strings through the global heap, IEEE floats and integers), contiguous / compact / chunked (v1 chunk B-tree; v4 single chunk)
 *  layouts and the deflate, shuffle and fletcher32 filters -- i.e. what MinKNOW's and this repo's writers produce.
 *
 *  It is a FAST PATH, not a second HDF5 library: anything it does not know (dense link / attribute storage, shared messages, other
 *  chunk indices, other filters such as VBZ, a Signal that is not little-endian int16, a missing attribute, any out-of-range
 *  address) makes fast5_read_raw_fast return 0 with nothing allocated, and read_raw (fast5_interface.c) then does what it always
 *  did through libhdf5 -- same values, same warnings.  Every access is bounds-checked against the bytes read.
 *  Format: "HDF5 File Format Specification Version 3.0" (the HDF Group); nothing here is taken from libhdf5's source.
 *  tests/test_fast5_raw.py holds it to the libhdf5 path on every layout fast5_tool can write.
 */
#include <fcntl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>
#include "fast5_raw.h"

#define UNDEF_ADDR 0xffffffffffffffffull
#define MAX_MSG 256
#define MAX_DEPTH 8

typedef struct { const unsigned char *p; size_t n; uint64_t base; } h5file;
typedef struct { unsigned type, flags; const unsigned char *data; size_t size; } h5msg;
typedef struct { h5msg m[MAX_MSG]; int n; } h5obj;

static uint64_t le(const unsigned char *p, int nbytes) {
    uint64_t v = 0;
    for (int i = nbytes - 1; i >= 0; i--) v = (v << 8) | p[i];
    return v;
}
/* pointer to `len` bytes at file address `addr` (relative to the base address), or NULL */
static const unsigned char *at(const h5file *f, uint64_t addr, uint64_t len) {
    if (UNDEF_ADDR == addr || addr > f->n || f->base > f->n - addr) return NULL;
    const uint64_t a = addr + f->base;
    if (len > f->n - a) return NULL;
    return f->p + a;
}

/* ---- object headers ---------------------------------------------------------------------------------------------------------- */
static int add_msg(h5obj *o, unsigned type, unsigned flags, const unsigned char *data, size_t size) {
    if (o->n >= MAX_MSG) return 0;
    o->m[o->n++] = (h5msg){ type, flags, data, size };
    return 1;
}

/* the messages of one block; continuation messages (type 0x10) are followed as they are met.  v2: 1-byte type, optional creation order */
static int parse_block(const h5file *f, h5obj *o, const unsigned char *p, size_t len, int v2, int corder, int depth) {
    if (depth > 32) return 0;
    const size_t hdr = v2 ? (size_t)(4 + (corder ? 2 : 0)) : 8;
    size_t off = 0;
    while (off + hdr <= len) {
        unsigned type, flags; size_t size;
        if (v2) { type = p[off]; size = (size_t)le(p + off + 1, 2); flags = p[off + 3]; }
        else    { type = (unsigned)le(p + off, 2); size = (size_t)le(p + off + 2, 2); flags = p[off + 4]; }
        off += hdr;
        if (size > len - off) return 0;
        const unsigned char *d = p + off;
        off += size;
        if (0x10 == type) {                                                      /* continuation: offset, length */
            if (size < 16) return 0;
            const uint64_t caddr = le(d, 8), clen = le(d + 8, 8);
            const unsigned char *c = at(f, caddr, clen);
            if (NULL == c) return 0;
            if (v2) {
                if (clen < 8 || 0 != memcmp(c, "OCHK", 4)) return 0;
                if (!parse_block(f, o, c + 4, (size_t)clen - 8, 1, corder, depth + 1)) return 0;
            } else if (!parse_block(f, o, c, (size_t)clen, 0, 0, depth + 1)) return 0;
        } else if (0 != type) {                                                  /* 0 = NIL */
            if (flags & 2) return 0;                                             /* shared message: not here */
            if (!add_msg(o, type, flags, d, size)) return 0;
        }
    }
    return 1;
}

static int read_object(const h5file *f, uint64_t addr, h5obj *o) {
    o->n = 0;
    const unsigned char *p = at(f, addr, 16);
    if (NULL == p) return 0;
    if (0 == memcmp(p, "OHDR", 4)) {
        if (2 != p[4]) return 0;
        const unsigned fl = p[5];
        size_t off = 6;
        if (fl & 0x20) off += 16;                                                /* four time stamps */
        if (fl & 0x10) off += 4;                                                 /* max compact / min dense attributes */
        const int szb = 1 << (fl & 3);
        const unsigned char *q = at(f, addr, off + (size_t)szb);
        if (NULL == q) return 0;
        const uint64_t c0 = le(q + off, szb);
        off += (size_t)szb;
        const unsigned char *blk = at(f, addr + off, c0 + 4);                    /* + checksum */
        if (NULL == blk) return 0;
        return parse_block(f, o, blk, (size_t)c0, 1, (fl & 4) != 0, 0);
    }
    if (1 != p[0]) return 0;                                                     /* version 1: 12 bytes + 4 of padding */
    const uint64_t hsize = le(p + 8, 4);
    const unsigned char *blk = at(f, addr + 16, hsize);
    if (NULL == blk) return 0;
    return parse_block(f, o, blk, (size_t)hsize, 0, 0, 0);
}

static const h5msg *find_msg(const h5obj *o, unsigned type) {
    for (int i = 0; i < o->n; i++) if (o->m[i].type == type) return &o->m[i];
    return NULL;
}

/* ---- groups ------------------------------------------------------------------------------------------------------------------- */
/* visit(name, address) for every link of a group; want == NULL: keep the smallest name in strcmp order (index 0 of H5_INDEX_NAME, increasing) */
typedef struct { const char *want; char best[256]; uint64_t addr; int found; } pick;

static int offer(pick *pk, const char *name, size_t len, uint64_t addr) {
    if (len >= sizeof(pk->best)) return 0;                                       /* a name this long: not compared here */
    if (NULL != pk->want) {
        if (strlen(pk->want) == len && 0 == memcmp(pk->want, name, len)) { pk->addr = addr; pk->found = 1; }
        return 1;
    }
    char tmp[256];
    memcpy(tmp, name, len); tmp[len] = 0;
    if (!pk->found || strcmp(tmp, pk->best) < 0) { memcpy(pk->best, tmp, len + 1); pk->addr = addr; pk->found = 1; }
    return 1;
}

static int walk_group_btree(const h5file *f, uint64_t addr, const unsigned char *heap, size_t heap_len, pick *pk, int depth) {
    if (depth > MAX_DEPTH) return 0;
    const unsigned char *p = at(f, addr, 24);
    if (NULL == p || 0 != memcmp(p, "TREE", 4) || 0 != p[4]) return 0;
    const unsigned level = p[5], n = (unsigned)le(p + 6, 2);
    const unsigned char *e = at(f, addr + 24, (uint64_t)n * 16 + 8);             /* key0, child0, key1, ... key n */
    if (NULL == e) return 0;
    for (unsigned i = 0; i < n; i++) {
        const uint64_t child = le(e + 8 + (size_t)i * 16, 8);
        if (level > 0) { if (!walk_group_btree(f, child, heap, heap_len, pk, depth + 1)) return 0; continue; }
        const unsigned char *s = at(f, child, 8);
        if (NULL == s || 0 != memcmp(s, "SNOD", 4)) return 0;
        const unsigned nsym = (unsigned)le(s + 6, 2);
        const unsigned char *ent = at(f, child + 8, (uint64_t)nsym * 40);
        if (NULL == ent) return 0;
        for (unsigned k = 0; k < nsym; k++) {
            const uint64_t noff = le(ent + (size_t)k * 40, 8), oaddr = le(ent + (size_t)k * 40 + 8, 8);
            if (noff >= heap_len) return 0;
            const void *z = memchr(heap + noff, 0, heap_len - (size_t)noff);
            if (NULL == z) return 0;
            if (!offer(pk, (const char *)heap + noff, (size_t)((const unsigned char *)z - (heap + noff)), oaddr)) return 0;
        }
    }
    return 1;
}

/* 1 = looked everywhere the group keeps links (pk->found says whether the name is there), 0 = a form this file does not read */
static int group_lookup(const h5file *f, uint64_t gaddr, pick *pk) {
    h5obj o;
    if (!read_object(f, gaddr, &o)) return 0;
    int seen = 0;
    const h5msg *st = find_msg(&o, 0x11);                                        /* symbol table: B-tree address, local heap address */
    if (NULL != st) {
        if (st->size < 16) return 0;
        const unsigned char *h = at(f, le(st->data + 8, 8), 32);
        if (NULL == h || 0 != memcmp(h, "HEAP", 4)) return 0;
        const uint64_t hsize = le(h + 8, 8);
        const unsigned char *seg = at(f, le(h + 24, 8), hsize);
        if (NULL == seg) return 0;
        if (!walk_group_btree(f, le(st->data, 8), seg, (size_t)hsize, pk, 0)) return 0;
        seen = 1;
    }
    const h5msg *li = find_msg(&o, 0x02);                                        /* link info: dense storage if its heap address is defined */
    if (NULL != li) {
        if (li->size < 2) return 0;
        size_t off = 2 + ((li->data[1] & 1) ? 8 : 0);
        if (li->size < off + 16) return 0;
        if (UNDEF_ADDR != le(li->data + off, 8)) return 0;
        seen = 1;
    }
    for (int i = 0; i < o.n; i++) {                                              /* link messages */
        if (0x06 != o.m[i].type) continue;
        const unsigned char *d = o.m[i].data; const size_t sz = o.m[i].size;
        if (sz < 2 || 1 != d[0]) return 0;
        const unsigned fl = d[1];
        size_t off = 2;
        unsigned ltype = 0;
        if (fl & 0x08) { if (off >= sz) return 0; ltype = d[off++]; }
        if (fl & 0x04) off += 8;                                                 /* creation order */
        if (fl & 0x10) off += 1;                                                 /* character set */
        const int lb = 1 << (fl & 3);
        if (off + (size_t)lb > sz) return 0;
        const uint64_t nlen = le(d + off, lb);
        off += (size_t)lb;
        if (nlen > sz - off) return 0;
        const char *name = (const char *)d + off;
        off += (size_t)nlen;
        if (0 != ltype) return 0;                                                /* soft / external links: libhdf5 resolves those */
        if (off + 8 > sz) return 0;
        if (!offer(pk, name, (size_t)nlen, le(d + off, 8))) return 0;
        seen = 1;
    }
    return seen;
}

static int child(const h5file *f, uint64_t gaddr, const char *name, uint64_t *out) {
    pick pk; memset(&pk, 0, sizeof(pk)); pk.want = name;
    if (!group_lookup(f, gaddr, &pk) || !pk.found) return 0;
    *out = pk.addr;
    return 1;
}

/* ---- datatypes, dataspaces, attributes ------------------------------------------------------------------------------------- */
enum { K_INT = 1, K_FLOAT, K_STR, K_VSTR };
typedef struct { int kind, size, is_signed; } h5type;

static int parse_type(const unsigned char *d, size_t sz, h5type *t) {
    if (sz < 8) return 0;
    const unsigned cls = d[0] & 0x0f, b0 = d[1];
    const uint64_t size = le(d + 4, 4);
    if (size < 1 || size > (1u << 20)) return 0;
    t->size = (int)size; t->is_signed = 0;
    switch (cls) {
    case 0:                                                                      /* fixed point: little-endian, whole bytes */
        if (sz < 12 || (b0 & 1) || 0 != le(d + 8, 2) || 8 * size != le(d + 10, 2) || (1 != size && 2 != size && 4 != size && 8 != size)) return 0;
        t->kind = K_INT; t->is_signed = (b0 >> 3) & 1;
        return 1;
    case 1: {                                                                    /* floating point: IEEE binary32 / binary64, little-endian */
        if (sz < 20 || (b0 & 0x41)) return 0;
        const unsigned boff = (unsigned)le(d + 8, 2), prec = (unsigned)le(d + 10, 2), eloc = d[12], esz = d[13], mloc = d[14], msz = d[15];
        const uint64_t bias = le(d + 16, 4);
        if (4 == size && 0 == boff && 32 == prec && 23 == eloc && 8 == esz && 0 == mloc && 23 == msz && 127 == bias) { t->kind = K_FLOAT; return 1; }
        if (8 == size && 0 == boff && 64 == prec && 52 == eloc && 11 == esz && 0 == mloc && 52 == msz && 1023 == bias) { t->kind = K_FLOAT; return 1; }
        return 0;
    }
    case 3: t->kind = K_STR; return 1;
    case 9:                                                                      /* variable length: strings only */
        if (1 != (b0 & 0x0f)) return 0;
        t->kind = K_VSTR; return 1;
    default: return 0;
    }
}

/* number of elements of a dataspace message, rank <= 1 (scalar = 1); -1 = not read here */
static int64_t space_count(const unsigned char *d, size_t sz, int *rank) {
    if (sz < 4) return -1;
    const unsigned ver = d[0], rk = d[1];
    size_t off;
    if (1 == ver) off = 8;
    else if (2 == ver) { off = 4; if (2 == d[3]) return -1; }                    /* null dataspace */
    else return -1;
    *rank = (int)rk;
    if (0 == rk) return 1;
    if (1 != rk || sz < off + 8) return -1;
    const uint64_t n = le(d + off, 8);
    return n > (1ull << 40) ? -1 : (int64_t)n;
}

typedef struct { h5type type; int64_t count; const unsigned char *data; size_t size; } h5attr;

/* 1 = found, 0 = not among the object's attribute messages or in a form not read here */
static int find_attr(const h5obj *o, const char *name, h5attr *a) {
    const size_t want = strlen(name);
    for (int i = 0; i < o->n; i++) {
        if (0x0c != o->m[i].type) continue;
        const unsigned char *d = o->m[i].data; const size_t sz = o->m[i].size;
        if (sz < 8) return 0;
        const unsigned ver = d[0];
        if (ver < 1 || ver > 3) return 0;
        if (ver > 1 && (d[1] & 3)) return 0;                                     /* shared datatype / dataspace */
        const size_t nsz = (size_t)le(d + 2, 2), tsz = (size_t)le(d + 4, 2), ssz = (size_t)le(d + 6, 2);
        size_t off = 3 == ver ? 9 : 8;
        const size_t pad = 1 == ver ? 7 : 0;
        const size_t n_p = (nsz + pad) & ~pad, t_p = (tsz + pad) & ~pad, s_p = (ssz + pad) & ~pad;
        if (off > sz || n_p > sz - off || t_p > sz - off - n_p || s_p > sz - off - n_p - t_p) return 0;
        const char *nm = (const char *)d + off;
        const size_t nlen = strnlen(nm, nsz);
        if (nlen != want || 0 != memcmp(nm, name, want)) continue;
        off += n_p;
        if (!parse_type(d + off, tsz, &a->type)) return 0;
        off += t_p;
        int rank = 0;
        a->count = space_count(d + off, ssz, &rank);
        if (a->count < 0) return 0;
        off += s_p;
        a->data = d + off; a->size = sz - off;
        return 1;
    }
    return 0;
}

static int attr_float(const h5obj *o, const char *name, float *out) {
    h5attr a;
    if (!find_attr(o, name, &a) || 1 != a.count || a.size < (size_t)a.type.size) return 0;
    if (K_FLOAT == a.type.kind) {
        if (4 == a.type.size) { float v; memcpy(&v, a.data, 4); *out = v; }
        else { double v; memcpy(&v, a.data, 8); *out = (float)v; }
        return 1;
    }
    if (K_INT == a.type.kind) {
        const uint64_t u = le(a.data, a.type.size);
        if (a.type.is_signed) {
            const int sh = 64 - 8 * a.type.size;
            *out = (float)((int64_t)(u << sh) >> sh);
        } else *out = (float)u;
        return 1;
    }
    return 0;
}

/* fixed string: the stored bytes + a terminator (what H5Aread into calloc(size + 1) leaves, fast5_interface.c:181-196);
 * variable length: the heap object's bytes up to its first NUL */
static char *attr_string(const h5file *f, const h5obj *o, const char *name) {
    h5attr a;
    if (!find_attr(o, name, &a) || 1 != a.count) return NULL;
    if (K_STR == a.type.kind) {
        if (a.size < (size_t)a.type.size) return NULL;
        char *s = calloc((size_t)a.type.size + 1, 1);
        if (NULL != s) memcpy(s, a.data, (size_t)a.type.size);
        return s;
    }
    if (K_VSTR != a.type.kind || a.size < 16) return NULL;
    const uint64_t len = le(a.data, 4), gaddr = le(a.data + 4, 8), idx = le(a.data + 12, 4);
    const unsigned char *g = at(f, gaddr, 16);
    if (NULL == g || 0 != memcmp(g, "GCOL", 4) || 1 != g[4]) return NULL;
    const uint64_t csize = le(g + 8, 8);
    const unsigned char *c = at(f, gaddr, csize);
    if (NULL == c || csize < 16) return NULL;
    for (uint64_t off = 16; off + 16 <= csize; ) {
        const uint64_t oi = le(c + off, 2), osz = le(c + off + 8, 8);
        if (0 == oi) break;                                                      /* free space: the collection's last object */
        if (osz > csize - off - 16) return NULL;
        if (oi == idx) {
            if (len > osz) return NULL;
            const size_t l = strnlen((const char *)c + off + 16, (size_t)len);
            char *s = malloc(l + 1);
            if (NULL != s) { memcpy(s, c + off + 16, l); s[l] = 0; }
            return s;
        }
        off += 16 + ((osz + 7) & ~7ull);
    }
    return NULL;
}

/* ---- the Signal dataset ------------------------------------------------------------------------------------------------------ */
typedef struct { int n; struct { unsigned id; unsigned cd0; } f[8]; } h5filters;

static int parse_filters(const h5msg *m, h5filters *fl) {
    fl->n = 0;
    if (NULL == m) return 1;
    const unsigned char *d = m->data; const size_t sz = m->size;
    if (sz < 2) return 0;
    const unsigned ver = d[0], nf = d[1];
    if ((1 != ver && 2 != ver) || nf > 8) return 0;
    size_t off = 1 == ver ? 8 : 2;
    for (unsigned i = 0; i < nf; i++) {
        if (off + 8 > sz) return 0;
        const unsigned id = (unsigned)le(d + off, 2);
        size_t nlen = 0;
        off += 2;
        if (1 == ver || id >= 256) { nlen = (size_t)le(d + off, 2); off += 2; }
        if (off + 4 > sz) return 0;
        const unsigned ncd = (unsigned)le(d + off + 2, 2);
        off += 4;
        if (1 == ver) nlen = (nlen + 7) & ~(size_t)7;
        if (nlen > sz - off) return 0;
        off += nlen;
        if ((size_t)ncd * 4 > sz - off) return 0;
        fl->f[i].id = id;
        fl->f[i].cd0 = ncd > 0 ? (unsigned)le(d + off, 4) : 0;
        off += (size_t)ncd * 4;
        if (1 == ver && (ncd & 1)) off += 4;
        if (1 != id && 2 != id && 3 != id) return 0;                             /* deflate, shuffle, fletcher32 */
    }
    fl->n = (int)nf;
    return 1;
}

/* one stored chunk -> `want` bytes of elements at dst (the filters undone last to first; mask bit i = filter i was skipped) */
static int unfilter(const h5filters *fl, unsigned mask, const unsigned char *src, size_t len, unsigned char *dst, size_t want, size_t esize, unsigned char *tmp) {
    const unsigned char *cur = src; size_t cl = len;
    for (int i = fl->n - 1; i >= 0; i--) {
        if (mask & (1u << i)) continue;
        unsigned char *out = (cur == dst) ? tmp : dst;
        if (1 == fl->f[i].id) {                                                  /* deflate */
            uLongf dl = (uLongf)want;
            if (Z_OK != uncompress(out, &dl, cur, (uLong)cl) || dl != want) return 0;
            cur = out; cl = want;
        } else if (2 == fl->f[i].id) {                                           /* shuffle: byte j of every element together */
            if (cl != want || 0 != want % esize) return 0;
            const size_t ne = want / esize;
            for (size_t j = 0; j < esize; j++) for (size_t e = 0; e < ne; e++) out[e * esize + j] = cur[j * ne + e];
            cur = out;
        } else if (3 == fl->f[i].id) {                                           /* fletcher32: a checksum behind the data, not verified */
            if (cl < 4) return 0;
            cl -= 4;
        } else return 0;
    }
    if (cl != want) return 0;
    if (cur != dst) memcpy(dst, cur, want);
    return 1;
}

static int walk_chunk_btree(const h5file *f, uint64_t addr, int ndim, const h5filters *fl, uint64_t chunk_elems, uint64_t n, int16_t *dst,
                            unsigned char *cbuf, unsigned char *tmp, uint64_t *covered, int depth) {
    if (depth > MAX_DEPTH) return 0;
    const unsigned char *p = at(f, addr, 24);
    if (NULL == p || 0 != memcmp(p, "TREE", 4) || 1 != p[4]) return 0;
    const unsigned level = p[5], ne = (unsigned)le(p + 6, 2);
    const size_t ksz = 8 + 8 * (size_t)ndim;
    const unsigned char *e = at(f, addr + 24, (uint64_t)ne * (ksz + 8) + ksz);
    if (NULL == e) return 0;
    for (unsigned i = 0; i < ne; i++) {
        const unsigned char *k = e + (size_t)i * (ksz + 8);
        const uint64_t csize = le(k, 4), off0 = le(k + 8, 8), caddr = le(k + ksz, 8);
        const unsigned mask = (unsigned)le(k + 4, 4);
        if (level > 0) { if (!walk_chunk_btree(f, caddr, ndim, fl, chunk_elems, n, dst, cbuf, tmp, covered, depth + 1)) return 0; continue; }
        if (off0 >= n || 0 != off0 % chunk_elems) return 0;
        const unsigned char *c = at(f, caddr, csize);
        if (NULL == c) return 0;
        if (!unfilter(fl, mask, c, (size_t)csize, cbuf, (size_t)chunk_elems * 2, 2, tmp)) return 0;
        const uint64_t take = n - off0 < chunk_elems ? n - off0 : chunk_elems;
        memcpy(dst + off0, cbuf, (size_t)take * 2);
        *covered += take;
    }
    return 1;
}

/* Signal as little-endian int16 -> malloc'd shorts; 0 = not read here */
static int read_signal(const h5file *f, uint64_t daddr, int16_t **out, uint64_t *nout) {
    h5obj o;
    if (!read_object(f, daddr, &o)) return 0;
    const h5msg *mt = find_msg(&o, 0x03), *ms = find_msg(&o, 0x01), *ml = find_msg(&o, 0x08);
    if (NULL == mt || NULL == ms || NULL == ml) return 0;
    h5type t;
    if (!parse_type(mt->data, mt->size, &t) || K_INT != t.kind || 2 != t.size || !t.is_signed) return 0;
    int rank = 0;
    const int64_t cnt = space_count(ms->data, ms->size, &rank);
    if (cnt <= 0 || 1 != rank) return 0;
    const uint64_t n = (uint64_t)cnt;
    const uint64_t most = (uint64_t)f->n * 516 + 65536;                         /* elements a file of this size can hold: deflate expands by at most ~1032 */
    if (n > most) return 0;
    h5filters fl;
    if (!parse_filters(find_msg(&o, 0x0b), &fl)) return 0;
    const unsigned char *d = ml->data; const size_t sz = ml->size;
    if (sz < 2 || (3 != d[0] && 4 != d[0])) return 0;
    int16_t *sig = malloc((size_t)n * 2);
    if (NULL == sig) return 0;
    int ok = 0;
    if (1 == d[1] && sz >= 18) {                                                 /* contiguous */
        const unsigned char *src = at(f, le(d + 2, 8), n * 2);
        if (NULL != src && le(d + 10, 8) >= n * 2 && 0 == fl.n) { memcpy(sig, src, (size_t)n * 2); ok = 1; }
    } else if (0 == d[1] && sz >= 4) {                                           /* compact */
        const uint64_t csz = le(d + 2, 2);
        if (csz >= n * 2 && sz >= 4 + n * 2 && 0 == fl.n) { memcpy(sig, d + 4, (size_t)n * 2); ok = 1; }
    } else if (2 == d[1] && 3 == d[0] && sz >= 11) {                             /* chunked, v1 B-tree index */
        const int ndim = d[2];
        if (2 == ndim && sz >= 11 + 8) {
            const uint64_t bt = le(d + 3, 8), ce = le(d + 11, 4), es = le(d + 15, 4);
            if (2 == es && ce >= 1 && ce <= most) {
                unsigned char *cbuf = malloc((size_t)ce * 2), *tmp = malloc((size_t)ce * 2);
                uint64_t covered = 0;
                if (NULL != cbuf && NULL != tmp && walk_chunk_btree(f, bt, ndim, &fl, ce, n, sig, cbuf, tmp, &covered, 0) && covered == n) ok = 1;
                free(cbuf); free(tmp);
            }
        }
    } else if (2 == d[1] && 4 == d[0] && sz >= 5) {                              /* chunked, version 4: the single-chunk index only */
        const unsigned lfl = d[2], ndim = d[3], enc = d[4];
        size_t off = 5;
        if (2 == ndim && enc >= 1 && enc <= 8 && sz >= off + 2 * (size_t)enc + 1) {
            const uint64_t ce = le(d + off, (int)enc), es = le(d + off + enc, (int)enc);
            off += 2 * (size_t)enc;
            const unsigned itype = d[off++];
            if (1 == itype && 2 == es && ce >= n && ce <= most) {
                uint64_t csize = ce * 2; unsigned mask = 0;
                if (lfl & 2) { if (sz < off + 12) goto done; csize = le(d + off, 8); mask = (unsigned)le(d + off + 8, 4); off += 12; }
                if (sz >= off + 8) {
                    const unsigned char *c = at(f, le(d + off, 8), csize);
                    unsigned char *cbuf = malloc((size_t)ce * 2), *tmp = malloc((size_t)ce * 2);
                    if (NULL != c && NULL != cbuf && NULL != tmp && ((lfl & 2) || 0 == fl.n) && unfilter(&fl, mask, c, (size_t)csize, cbuf, (size_t)ce * 2, 2, tmp)) {
                        memcpy(sig, cbuf, (size_t)n * 2); ok = 1;
                    }
                    free(cbuf); free(tmp);
                }
            }
        }
    }
done:
    if (!ok) { free(sig); return 0; }
    *out = sig; *nout = n;
    return 1;
}

/* ---- the file ------------------------------------------------------------------------------------------------------------------ */
static unsigned char *slurp(const char *filename, size_t *len) {
    const int fd = open(filename, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return NULL;
    struct stat st;
    unsigned char *buf = NULL;
    if (0 == fstat(fd, &st) && S_ISREG(st.st_mode) && st.st_size >= 64 && st.st_size < ((off_t)1 << 31) && NULL != (buf = malloc((size_t)st.st_size))) {
        size_t got = 0;
        while (got < (size_t)st.st_size) {
            const ssize_t r = read(fd, buf + got, (size_t)st.st_size - got);
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got != (size_t)st.st_size) { free(buf); buf = NULL; }
        *len = got;
    }
    close(fd);
    return buf;
}

static int root_address(h5file *f, uint64_t *root) {
    static const unsigned char sig[8] = { 0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n' };
    if (f->n < 96 || 0 != memcmp(f->p, sig, 8)) return 0;                       /* a superblock behind a user block: left to libhdf5 */
    const unsigned ver = f->p[8];
    if (ver <= 1) {
        if (8 != f->p[13] || 8 != f->p[14]) return 0;                           /* sizes of offsets and lengths */
        const size_t o = 0 == ver ? 24 : 28;
        f->base = le(f->p + o, 8);
        *root = le(f->p + o + 32 + 8, 8);                                        /* root group's symbol table entry: name offset, header address */
        return 1;
    }
    if (ver <= 3) {
        if (8 != f->p[9] || 8 != f->p[10]) return 0;
        f->base = le(f->p + 12, 8);
        *root = le(f->p + 36, 8);
        return 1;
    }
    return 0;
}

int fast5_read_raw_fast(const char *filename, int scale_to_pA, fast5_raw_read *out) {
    memset(out, 0, sizeof(*out));
    size_t len = 0;
    unsigned char *buf = slurp(filename, &len);
    if (NULL == buf) return 0;
    h5file f = { buf, len, 0 };
    uint64_t root, g, reads, rd, ds;
    int16_t *sig = NULL; uint64_t n = 0;
    char *uuid = NULL;
    int ok = 0;
    h5obj *o = malloc(sizeof(h5obj));
    if (NULL == o || !root_address(&f, &root)) goto out;
    if (!child(&f, root, "Raw", &g) || !child(&f, g, "Reads", &reads)) goto out;
    {   pick pk; memset(&pk, 0, sizeof(pk));
        if (!group_lookup(&f, reads, &pk) || !pk.found) goto out;
        rd = pk.addr;
    }
    if (!read_object(&f, rd, o)) goto out;
    uuid = attr_string(&f, o, "read_id");
    if (NULL == uuid) goto out;
    if (!child(&f, rd, "Signal", &ds) || !read_signal(&f, ds, &sig, &n)) goto out;
    float digitisation = 0.0f, offset = 0.0f, range = 0.0f;
    if (scale_to_pA) {
        uint64_t u, c;
        if (!child(&f, root, "UniqueGlobalKey", &u) || !child(&f, u, "channel_id", &c) || !read_object(&f, c, o)) goto out;
        if (!attr_float(o, "digitisation", &digitisation) || !attr_float(o, "offset", &offset) || !attr_float(o, "range", &range)) goto out;
    }
    float *raw = malloc((size_t)n * sizeof(float));
    if (NULL == raw) goto out;
    for (uint64_t i = 0; i < n; i++) raw[i] = (float)sig[i];
    if (scale_to_pA) {                                                           /* fast5_interface.c:297-303 */
        const float raw_unit = range / digitisation;
        for (uint64_t i = 0; i < n; i++) raw[i] = (raw[i] + offset) * raw_unit;
    }
    out->uuid = uuid; uuid = NULL;
    out->raw = raw; out->n = (size_t)n;
    ok = 1;
out:
    free(o); free(sig); free(uuid); free(buf);
    return ok;
}
