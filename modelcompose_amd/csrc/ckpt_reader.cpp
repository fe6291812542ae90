// Native checkpoint reader: the files load_pretrained_model reads with torch.load / safetensors in the reference
// (modelcompose/model/builder.py:157-168: adapter_model.bin, non_lora_trainables.bin; :148 base shards pytorch_model-0000x-of-0000y.bin;
// encoder checkpoints) - opened, indexed and handed out as (name, dtype, shape, strides, pointer into the mapped file) without Python
// unpickling anything and without a second host copy; tensors go from the page cache straight to HBM.
//
// Formats:
//   * torch zip checkpoints (torch.save since 1.6): a STORED (uncompressed) ZIP / ZIP64 archive holding `<root>/data.pkl` (a protocol-2
//     pickle of the object tree, storages referenced by persistent id) and one raw record `<root>/data/<key>` per storage.  The pickle is
//     interpreted by a small restricted stack machine: only the opcodes torch.save emits, only the globals that rebuild tensors /
//     parameters / ordered dicts (torch._utils._rebuild_tensor_v2, _rebuild_parameter, collections.OrderedDict, torch.<T>Storage);
//     any other global is kept as an opaque object and never called - a checkpoint cannot execute code here.
//   * safetensors: u64 header length, JSON header {name: {dtype, shape, data_offsets}}, raw bytes.
// Nested containers are flattened with '.'-joined keys (state dicts are flat in every file the path reads).
#include <errno.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mc_hip.h"

void mc_set_error(const char* fmt, ...);

namespace {

// Paths: the levels of the object tree joined by PATH_SEP; a level that is a list / tuple index starts with PATH_IDX (so a rebuilt tree
// can tell {"0": x} from [x]).  Names ('.'-joined) stay what the flat state-dict readers use.
constexpr char PATH_SEP = '\x1f', PATH_IDX = '\x1e';

struct Scalar {                     // non-tensor leaf of the object tree (BEATs checkpoints keep their config dict next to the weights)
    std::string name, path, s;
    int kind = MC_CKPT_NONE;        // MC_CKPT_NONE / _BOOLEAN / _INT / _FLOAT / _STR
    int64_t i = 0;
    double f = 0;
};

struct Entry {
    std::string name, path;
    int dtype = -1;                 // MC_CKPT_* code
    std::vector<int64_t> shape, stride;
    const uint8_t* data = nullptr;  // first element (storage base + storage_offset)
    int64_t storage_bytes_left = 0; // bytes from `data` to the end of its storage record
};

struct Ckpt {
    int fd = -1;
    uint8_t* map = nullptr;
    size_t size = 0;
    std::vector<Entry> entries;
    std::vector<Scalar> scalars;
    int64_t visited_nodes = 0;      // flatten()'s global budget
    ~Ckpt() {
        if (map) munmap(map, size);
        if (fd >= 0) close(fd);
    }
};

int elt_size(int dt) {
    switch (dt) {
        case MC_CKPT_F32: case MC_CKPT_I32: return 4;
        case MC_CKPT_F16: case MC_CKPT_BF16: case MC_CKPT_I16: return 2;
        case MC_CKPT_F64: case MC_CKPT_I64: return 8;
        case MC_CKPT_I8: case MC_CKPT_U8: case MC_CKPT_BOOL: return 1;
    }
    return 0;
}

// ------------------------------------------------------------------------------------------------ zip
uint16_t rd16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }
uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
uint64_t rd64(const uint8_t* p) { return (uint64_t)rd32(p) | ((uint64_t)rd32(p + 4) << 32); }

struct ZipRec { uint64_t off = 0, size = 0; };

// name -> (data offset, size) of every STORED record
bool parse_zip(const Ckpt& c, std::map<std::string, ZipRec>& recs, std::string& err) {
    const uint8_t* m = c.map;
    const size_t n = c.size;
    if (n < 22) { err = "file too small for a zip archive"; return false; }
    // end-of-central-directory record: scan back over a possible archive comment
    size_t eocd = (size_t)-1;
    const size_t lo = n > 22 + 65535 ? n - 22 - 65535 : 0;
    for (size_t i = n - 22 + 1; i-- > lo;)
        if (rd32(m + i) == 0x06054b50u) { eocd = i; break; }
    if (eocd == (size_t)-1) { err = "no zip end-of-central-directory record (legacy, non-zip torch checkpoints are not supported)"; return false; }
    uint64_t cd_off = rd32(m + eocd + 16), cd_size = rd32(m + eocd + 12), total = rd16(m + eocd + 10);
    if (eocd >= 20 && rd32(m + eocd - 20) == 0x07064b50u) {            // zip64 locator -> zip64 end-of-central-directory record
        const uint64_t e64 = rd64(m + eocd - 20 + 8);
        if (e64 > n || n - e64 < 56 || rd32(m + e64) != 0x06064b50u) { err = "corrupt zip64 end-of-central-directory record"; return false; }
        total = rd64(m + e64 + 32); cd_size = rd64(m + e64 + 40); cd_off = rd64(m + e64 + 48);
    }
    // every bound below is written as `x > n || y > n - x`: fields of an untrusted file must not wrap a 64-bit sum
    if (cd_off > n || cd_size > n - cd_off) { err = "central directory outside the file"; return false; }
    if (total > cd_size / 46) { err = "central directory entry count exceeds its size"; return false; }
    uint64_t p = cd_off;
    for (uint64_t k = 0; k < total; ++k) {
        if (p > n || n - p < 46 || rd32(m + p) != 0x02014b50u) { err = "corrupt central directory entry"; return false; }
        const uint16_t method = rd16(m + p + 10), nlen = rd16(m + p + 28), xlen = rd16(m + p + 30), clen = rd16(m + p + 32);
        uint64_t csize = rd32(m + p + 20), usize = rd32(m + p + 24), lho = rd32(m + p + 42);
        if ((uint64_t)nlen + xlen + clen > n - p - 46) { err = "corrupt central directory entry"; return false; }
        std::string name((const char*)m + p + 46, nlen);
        // zip64 extended information: 64-bit fields replace the 0xFFFFFFFF placeholders, in this fixed order
        uint64_t x = p + 46 + nlen;
        const uint64_t xend = x + xlen;
        while (x + 4 <= xend) {
            const uint16_t id = rd16(m + x), sz = rd16(m + x + 2);
            if (id == 0x0001) {
                uint64_t q = x + 4;
                if (usize == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { usize = rd64(m + q); q += 8; }
                if (csize == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { csize = rd64(m + q); q += 8; }
                if (lho == 0xFFFFFFFFu && q + 8 <= x + 4 + sz) { lho = rd64(m + q); q += 8; }
            }
            x += 4 + sz;
        }
        if (lho > n || n - lho < 30 || rd32(m + lho) != 0x04034b50u) { err = "corrupt local file header of '" + name + "'"; return false; }
        const uint64_t data = lho + 30 + rd16(m + lho + 26) + rd16(m + lho + 28);            // lho <= n - 30: cannot wrap
        if (method != 0 || csize != usize) { err = "record '" + name + "' is compressed; torch checkpoints are stored uncompressed"; return false; }
        if (data > n || usize > n - data) { err = "record '" + name + "' outside the file"; return false; }
        recs[name] = ZipRec{data, usize};
        p += 46 + nlen + xlen + clen;
    }
    return true;
}

// ------------------------------------------------------------------------------------------------ pickle (restricted)
struct Val;
using VP = std::shared_ptr<Val>;
struct Val {
    enum Kind { NONE, BOOL, INT, FLOAT, STR, TUPLE, LIST, DICT, GLOBAL, STORAGE, TENSOR, OBJECT, MARK } kind = NONE;
    int64_t i = 0;
    double f = 0;
    std::string s, s2;                       // STR: s; GLOBAL: module s, name s2; STORAGE: key s
    std::vector<VP> items;                   // TUPLE / LIST
    std::vector<std::pair<VP, VP>> dict;     // DICT (insertion order)
    int dtype = -1;                          // STORAGE / TENSOR
    int64_t numel = 0, offset = 0;           // STORAGE numel; TENSOR storage offset (elements)
    std::vector<int64_t> shape, stride;
    VP storage;
};
VP mk(Val::Kind k) { auto v = std::make_shared<Val>(); v->kind = k; return v; }

int storage_dtype(const std::string& name) {
    static const std::pair<const char*, int> tab[] = {
        {"FloatStorage", MC_CKPT_F32}, {"HalfStorage", MC_CKPT_F16}, {"BFloat16Storage", MC_CKPT_BF16}, {"DoubleStorage", MC_CKPT_F64},
        {"LongStorage", MC_CKPT_I64}, {"IntStorage", MC_CKPT_I32}, {"ShortStorage", MC_CKPT_I16}, {"CharStorage", MC_CKPT_I8},
        {"ByteStorage", MC_CKPT_U8}, {"BoolStorage", MC_CKPT_BOOL}};
    for (auto& t : tab)
        if (name == t.first) return t.second;
    return -1;
}

bool to_int_list(const VP& v, std::vector<int64_t>& out) {
    if (!v || (v->kind != Val::TUPLE && v->kind != Val::LIST)) return false;
    for (auto& e : v->items) {
        if (!e || e->kind != Val::INT) return false;
        out.push_back(e->i);
    }
    return true;
}

struct Unpickler {
    const uint8_t* p;
    const uint8_t* end;
    std::vector<VP> stack;
    std::map<uint32_t, VP> memo;
    std::string err;

    bool need(size_t n) {
        if ((size_t)(end - p) < n) { err = "truncated pickle"; return false; }
        return true;
    }
    bool pop(VP& v) {
        if (stack.empty() || stack.back()->kind == Val::MARK) { err = "pickle stack underflow"; return false; }
        v = stack.back(); stack.pop_back();
        return true;
    }
    bool pop_mark(std::vector<VP>& items) {
        size_t k = stack.size();
        while (k > 0 && stack[k - 1]->kind != Val::MARK) --k;
        if (k == 0) { err = "pickle MARK not found"; return false; }
        items.assign(stack.begin() + k, stack.end());
        stack.resize(k - 1);
        return true;
    }
    bool read_line(std::string& s) {
        const uint8_t* q = (const uint8_t*)memchr(p, '\n', end - p);
        if (!q) { err = "truncated pickle"; return false; }
        s.assign((const char*)p, q - p);
        p = q + 1;
        return true;
    }
    VP reduce(const VP& fn, const VP& args) {
        if (fn->kind == Val::GLOBAL && args->kind == Val::TUPLE) {
            const std::string full = fn->s + "." + fn->s2;
            if (full == "collections.OrderedDict" || full == "builtins.dict" || full == "__builtin__.dict") return mk(Val::DICT);
            if ((full == "torch._utils._rebuild_tensor_v2" || full == "torch._utils._rebuild_tensor") && args->items.size() >= 4 &&
                args->items[0]->kind == Val::STORAGE && args->items[1]->kind == Val::INT) {
                auto t = mk(Val::TENSOR);
                t->storage = args->items[0];
                t->dtype = t->storage->dtype;
                t->offset = args->items[1]->i;
                if (!to_int_list(args->items[2], t->shape) || !to_int_list(args->items[3], t->stride) || t->shape.size() != t->stride.size())
                    return mk(Val::OBJECT);
                return t;
            }
            if (full == "torch._utils._rebuild_parameter" && !args->items.empty() && args->items[0]->kind == Val::TENSOR) return args->items[0];
            if (full == "torch._utils._rebuild_parameter_with_state" && !args->items.empty() && args->items[0]->kind == Val::TENSOR) return args->items[0];
        }
        return mk(Val::OBJECT);          // anything else is inert: nothing in a checkpoint is ever called
    }
    VP persistent_load(const VP& pid) {
        // ('storage', <storage type global>, key, location, numel)
        if (pid->kind == Val::TUPLE && pid->items.size() >= 5 && pid->items[0]->kind == Val::STR && pid->items[0]->s == "storage" &&
            pid->items[1]->kind == Val::GLOBAL && pid->items[2]->kind == Val::STR && pid->items[4]->kind == Val::INT) {
            auto s = mk(Val::STORAGE);
            s->dtype = storage_dtype(pid->items[1]->s2);
            s->s = pid->items[2]->s;
            s->numel = pid->items[4]->i;
            return s;
        }
        return mk(Val::OBJECT);
    }
    bool run(VP& result) {
        while (p < end) {
            const uint8_t op = *p++;
            switch (op) {
                case 0x80: if (!need(1)) return false; ++p; break;                                   // PROTO
                case 0x95: if (!need(8)) return false; p += 8; break;                                // FRAME (protocol 4)
                case '.': return pop(result);                                                       // STOP
                case '(': stack.push_back(mk(Val::MARK)); break;
                case 'N': stack.push_back(mk(Val::NONE)); break;
                case 0x88: case 0x89: { auto v = mk(Val::BOOL); v->i = op == 0x88; stack.push_back(v); break; }
                case 'K': { if (!need(1)) return false; auto v = mk(Val::INT); v->i = *p++; stack.push_back(v); break; }
                case 'M': { if (!need(2)) return false; auto v = mk(Val::INT); v->i = rd16(p); p += 2; stack.push_back(v); break; }
                case 'J': { if (!need(4)) return false; auto v = mk(Val::INT); v->i = (int32_t)rd32(p); p += 4; stack.push_back(v); break; }
                case 0x8a: {                                                                        // LONG1
                    if (!need(1)) return false;
                    const int n = *p++;
                    if (!need(n) || n > 8) { err = "LONG1 wider than 8 bytes"; return false; }
                    int64_t v = 0;
                    for (int k = 0; k < n; ++k) v |= (int64_t)p[k] << (8 * k);
                    if (n > 0 && n < 8 && (p[n - 1] & 0x80)) v |= -((int64_t)1 << (8 * n));
                    p += n;
                    auto x = mk(Val::INT); x->i = v; stack.push_back(x);
                    break;
                }
                case 'G': {                                                                         // BINFLOAT (big endian)
                    if (!need(8)) return false;
                    uint64_t b = 0;
                    for (int k = 0; k < 8; ++k) b = (b << 8) | p[k];
                    p += 8;
                    auto v = mk(Val::FLOAT); memcpy(&v->f, &b, 8); stack.push_back(v);
                    break;
                }
                case 'X': case 0x8c: case 'U': case 'T': case 'B': case 'C': case 0x8d: {           // unicode / str / bytes with 1-, 4- or 8-byte length
                    size_t len;
                    if (op == 0x8c || op == 'U' || op == 'C') { if (!need(1)) return false; len = *p++; }
                    else if (op == 0x8d) { if (!need(8)) return false; len = (size_t)rd64(p); p += 8; }
                    else { if (!need(4)) return false; len = rd32(p); p += 4; }
                    if (!need(len)) return false;
                    auto v = mk(Val::STR); v->s.assign((const char*)p, len); p += len; stack.push_back(v);
                    break;
                }
                case 'c': {                                                                         // GLOBAL
                    auto v = mk(Val::GLOBAL);
                    if (!read_line(v->s) || !read_line(v->s2)) return false;
                    stack.push_back(v);
                    break;
                }
                case 0x93: {                                                                        // STACK_GLOBAL
                    VP name, mod;
                    if (!pop(name) || !pop(mod)) return false;
                    auto v = mk(Val::GLOBAL); v->s = mod->s; v->s2 = name->s; stack.push_back(v);
                    break;
                }
                case 'q': { if (!need(1) || stack.empty()) { err = "bad BINPUT"; return false; } memo[*p++] = stack.back(); break; }
                case 'r': { if (!need(4) || stack.empty()) { err = "bad LONG_BINPUT"; return false; } memo[rd32(p)] = stack.back(); p += 4; break; }
                case 0x94: { if (stack.empty()) { err = "bad MEMOIZE"; return false; } const uint32_t k = (uint32_t)memo.size(); memo[k] = stack.back(); break; }
                case 'h': case 'j': {
                    uint32_t k;
                    if (op == 'h') { if (!need(1)) return false; k = *p++; } else { if (!need(4)) return false; k = rd32(p); p += 4; }
                    auto it = memo.find(k);
                    if (it == memo.end()) { err = "pickle memo miss"; return false; }
                    stack.push_back(it->second);
                    break;
                }
                case ')': stack.push_back(mk(Val::TUPLE)); break;
                case '}': stack.push_back(mk(Val::DICT)); break;
                case ']': stack.push_back(mk(Val::LIST)); break;
                case 't': { auto v = mk(Val::TUPLE); if (!pop_mark(v->items)) return false; stack.push_back(v); break; }
                case 0x85: case 0x86: case 0x87: {
                    const int n = op - 0x84;
                    if ((int)stack.size() < n) { err = "pickle stack underflow"; return false; }
                    auto v = mk(Val::TUPLE);
                    v->items.assign(stack.end() - n, stack.end());
                    stack.resize(stack.size() - n);
                    stack.push_back(v);
                    break;
                }
                case 'a': { VP x, l; if (!pop(x)) return false; if (stack.empty()) { err = "APPEND on empty stack"; return false; } l = stack.back(); if (l->kind == Val::LIST) l->items.push_back(x); break; }
                case 'e': { std::vector<VP> xs; if (!pop_mark(xs) || stack.empty()) { err = "bad APPENDS"; return false; } if (stack.back()->kind == Val::LIST) for (auto& x : xs) stack.back()->items.push_back(x); break; }
                case 's': { VP v, k; if (!pop(v) || !pop(k) || stack.empty()) { err = "bad SETITEM"; return false; } if (stack.back()->kind == Val::DICT) stack.back()->dict.emplace_back(k, v); break; }
                case 'u': {
                    std::vector<VP> xs;
                    if (!pop_mark(xs) || stack.empty() || (xs.size() & 1)) { err = "bad SETITEMS"; return false; }
                    if (stack.back()->kind == Val::DICT)
                        for (size_t k = 0; k + 1 < xs.size(); k += 2) stack.back()->dict.emplace_back(xs[k], xs[k + 1]);
                    break;
                }
                case 'Q': { VP pid; if (!pop(pid)) return false; stack.push_back(persistent_load(pid)); break; }        // BINPERSID
                case 'R': { VP args, fn; if (!pop(args) || !pop(fn)) return false; stack.push_back(reduce(fn, args)); break; }
                case 0x81: { VP args, cls; if (!pop(args) || !pop(cls)) return false; stack.push_back(reduce(cls, args)); break; }   // NEWOBJ
                case 'b': {                                                                         // BUILD: state of an OrderedDict subclass etc.
                    VP state, obj;
                    if (!pop(state) || stack.empty()) { err = "bad BUILD"; return false; }
                    break;
                }
                default:
                    err = "unsupported pickle opcode 0x";
                    err += "0123456789abcdef"[op >> 4]; err += "0123456789abcdef"[op & 15];
                    return false;
            }
        }
        err = "pickle ended without STOP";
        return false;
    }
};

constexpr int MAX_DEPTH = 64;       // of the object tree (a pickle can build a self-referential container through its memo)
// of visited nodes: memo references make the tree a DAG, so `l = [l, l]` nested n times expands to 2^n leaves from a 300-byte file
// (ADVICE r3).  The largest real checkpoint on this path holds a few thousand leaves; 4 M is far above any of them.
constexpr int64_t MAX_NODES = 4 * 1000 * 1000;

bool flatten(const VP& v, const std::string& prefix, const std::string& path, int depth, Ckpt& c, const std::map<std::string, ZipRec>& recs,
             const std::string& root, std::string& err) {
    if (depth > MAX_DEPTH) { err = "object tree deeper than 64 levels (or self-referential) at '" + prefix + "'"; return false; }
    if (++c.visited_nodes > MAX_NODES) { err = "object tree expands to more than 4000000 nodes (shared containers referenced repeatedly?)"; return false; }
    auto join = [](const std::string& a, const std::string& b, char sep) { return a.empty() ? b : a + sep + b; };
    auto empty_marker = [&](int kind) {
        Scalar sc;
        sc.name = prefix; sc.path = path; sc.i = 0; sc.f = 0; sc.kind = kind;
        c.scalars.push_back(std::move(sc));
    };
    if (v->kind == Val::DICT) {
        if (v->dict.empty() && depth > 0) { empty_marker(MC_CKPT_EMPTY_DICT); return true; }
        for (auto& kv : v->dict) {
            std::string key;
            if (kv.first->kind == Val::STR) key = kv.first->s;
            else if (kv.first->kind == Val::INT) key = std::to_string(kv.first->i);
            else continue;
            if (!flatten(kv.second, join(prefix, key, '.'), join(path, key, PATH_SEP), depth + 1, c, recs, root, err)) return false;
        }
        return true;
    }
    if (v->kind == Val::LIST || v->kind == Val::TUPLE) {
        if (v->items.empty() && depth > 0) { empty_marker(v->kind == Val::LIST ? MC_CKPT_EMPTY_LIST : MC_CKPT_EMPTY_TUPLE); return true; }
        for (size_t k = 0; k < v->items.size(); ++k) {
            const std::string key = std::to_string(k);
            if (!flatten(v->items[k], join(prefix, key, '.'), join(path, std::string(1, PATH_IDX) + key, PATH_SEP), depth + 1, c, recs, root, err)) return false;
        }
        return true;
    }
    if (v->kind == Val::NONE || v->kind == Val::BOOL || v->kind == Val::INT || v->kind == Val::FLOAT || v->kind == Val::STR) {
        Scalar sc;
        sc.name = prefix; sc.path = path; sc.i = v->i; sc.f = v->f; sc.s = v->s;
        sc.kind = v->kind == Val::NONE ? MC_CKPT_NONE : v->kind == Val::BOOL ? MC_CKPT_BOOLEAN : v->kind == Val::INT ? MC_CKPT_INT
                : v->kind == Val::FLOAT ? MC_CKPT_FLOAT : MC_CKPT_STR;
        c.scalars.push_back(std::move(sc));
        return true;
    }
    if (v->kind != Val::TENSOR) return true;              // opaque objects carry nothing
    auto it = recs.find(root + "data/" + v->storage->s);
    if (it == recs.end()) { err = "storage record '" + v->storage->s + "' of tensor '" + prefix + "' is missing"; return false; }
    const int es = elt_size(v->dtype);
    if (es == 0) { err = "tensor '" + prefix + "' has an unsupported storage type"; return false; }
    // the furthest element a strided view touches must lie inside its storage record; every product / sum is overflow-checked
    // (shape, stride and offset come straight from the pickle)
    int64_t span = 1;
    bool empty = false;
    for (size_t d = 0; d < v->shape.size(); ++d) {
        if (v->shape[d] < 0 || v->stride[d] < 0) { err = "tensor '" + prefix + "' has a negative size or stride"; return false; }
        if (v->shape[d] == 0) empty = true;
    }
    const std::string outside = "tensor '" + prefix + "' reaches outside its storage record";
    if (!empty)
        for (size_t d = 0; d < v->shape.size(); ++d) {
            int64_t step;
            if (__builtin_mul_overflow(v->shape[d] - 1, v->stride[d], &step) || __builtin_add_overflow(span, step, &span)) { err = outside; return false; }
        }
    else span = 0;
    int64_t last, bytes, off_bytes;
    if (v->offset < 0 || __builtin_add_overflow(v->offset, span, &last) || __builtin_mul_overflow(last, (int64_t)es, &bytes) ||
        (uint64_t)bytes > it->second.size || __builtin_mul_overflow(v->offset, (int64_t)es, &off_bytes)) { err = outside; return false; }
    Entry e;
    e.name = prefix; e.path = path; e.dtype = v->dtype; e.shape = v->shape; e.stride = v->stride;
    e.data = c.map + it->second.off + (uint64_t)off_bytes;
    e.storage_bytes_left = (int64_t)(it->second.size - (uint64_t)off_bytes);
    c.entries.push_back(std::move(e));
    return true;
}

bool load_torch_zip(Ckpt& c, std::string& err) {
    std::map<std::string, ZipRec> recs;
    if (!parse_zip(c, recs, err)) return false;
    std::string root;
    const ZipRec* pkl = nullptr;
    for (auto& kv : recs) {
        const std::string& n = kv.first;
        if (n.size() >= 8 && n.compare(n.size() - 8, 8, "data.pkl") == 0 && (n.size() == 8 || n[n.size() - 9] == '/')) {
            root = n.substr(0, n.size() - 8);
            pkl = &kv.second;
            break;
        }
    }
    if (!pkl) { err = "zip archive holds no data.pkl (not a torch checkpoint)"; return false; }
    Unpickler u{c.map + pkl->off, c.map + pkl->off + pkl->size, {}, {}, {}};
    VP top;
    if (!u.run(top)) { err = "data.pkl: " + u.err; return false; }
    return flatten(top, "", "", 0, c, recs, root, err);
}

// ------------------------------------------------------------------------------------------------ safetensors
struct Json {
    const char* p;
    const char* end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool str(std::string& out) {
        ws();
        if (p >= end || *p != '"') { err = "expected a string"; return false; }
        ++p;
        out.clear();
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                switch (*p) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break; case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': {                     // \uXXXX: basic multilingual plane only (tensor names are ASCII in practice)
                        if (p + 4 >= end) { err = "bad \\u escape"; return false; }
                        unsigned cp = 0;
                        for (int k = 1; k <= 4; ++k) { const char ch = p[k]; cp = cp * 16 + (ch <= '9' ? ch - '0' : (ch | 32) - 'a' + 10); }
                        p += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 63)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 63)); out += (char)(0x80 | (cp & 63)); }
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else out += *p++;
        }
        if (p >= end) { err = "unterminated string"; return false; }
        ++p;
        return true;
    }
    bool skip(int depth = 0) {                      // any value
        if (depth > MAX_DEPTH) { err = "JSON nested deeper than 64 levels"; return false; }
        ws();
        if (p >= end) { err = "unexpected end"; return false; }
        if (*p == '"') { std::string s; return str(s); }
        if (*p == '{' || *p == '[') {
            const char close = *p == '{' ? '}' : ']';
            ++p; ws();
            if (p < end && *p == close) { ++p; return true; }
            while (true) {
                if (close == '}') { std::string k; if (!str(k)) return false; ws(); if (p >= end || *p != ':') { err = "expected ':'"; return false; } ++p; }
                if (!skip(depth + 1)) return false;
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == close) { ++p; return true; }
                err = "expected ',' or a closing bracket"; return false;
            }
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n') ++p;
        return true;
    }
    bool int_array(std::vector<int64_t>& out) {
        ws();
        if (p >= end || *p != '[') { err = "expected '['"; return false; }
        ++p; ws();
        if (p < end && *p == ']') { ++p; return true; }
        while (true) {
            ws();
            bool neg = false;
            if (p < end && *p == '-') { neg = true; ++p; }
            if (p >= end || *p < '0' || *p > '9') { err = "expected an integer"; return false; }
            int64_t v = 0;
            while (p < end && *p >= '0' && *p <= '9')
                if (__builtin_mul_overflow(v, (int64_t)10, &v) || __builtin_add_overflow(v, (int64_t)(*p++ - '0'), &v)) { err = "integer out of range"; return false; }
            out.push_back(neg ? -v : v);
            ws();
            if (p < end && *p == ',') { ++p; continue; }
            if (p < end && *p == ']') { ++p; return true; }
            err = "expected ',' or ']'"; return false;
        }
    }
};

int st_dtype(const std::string& s) {
    static const std::pair<const char*, int> tab[] = {{"F32", MC_CKPT_F32}, {"F16", MC_CKPT_F16}, {"BF16", MC_CKPT_BF16}, {"F64", MC_CKPT_F64},
                                                      {"I64", MC_CKPT_I64}, {"I32", MC_CKPT_I32}, {"I16", MC_CKPT_I16}, {"I8", MC_CKPT_I8},
                                                      {"U8", MC_CKPT_U8}, {"BOOL", MC_CKPT_BOOL}};
    for (auto& t : tab)
        if (s == t.first) return t.second;
    return -1;
}

bool load_safetensors(Ckpt& c, std::string& err) {
    if (c.size < 8) { err = "file too small"; return false; }
    const uint64_t hl = rd64(c.map);
    if (hl > c.size - 8) { err = "safetensors header length outside the file"; return false; }
    const uint8_t* base = c.map + 8 + hl;
    const uint64_t data_bytes = c.size - 8 - hl;
    Json j{(const char*)c.map + 8, (const char*)c.map + 8 + hl, {}};
    j.ws();
    if (j.p >= j.end || *j.p != '{') { err = "safetensors header is not a JSON object"; return false; }
    ++j.p; j.ws();
    if (j.p < j.end && *j.p == '}') return true;
    while (true) {
        std::string name;
        if (!j.str(name)) { err = "safetensors header: " + j.err; return false; }
        j.ws();
        if (j.p >= j.end || *j.p != ':') { err = "safetensors header: expected ':'"; return false; }
        ++j.p;
        if (name == "__metadata__") {
            if (!j.skip()) { err = "safetensors header: " + j.err; return false; }
        } else {
            j.ws();
            if (j.p >= j.end || *j.p != '{') { err = "safetensors header: tensor entry is not an object"; return false; }
            ++j.p;
            Entry e;
            e.name = name;
            std::vector<int64_t> offs;
            while (true) {
                std::string k;
                if (!j.str(k)) { err = "safetensors header: " + j.err; return false; }
                j.ws();
                if (j.p >= j.end || *j.p != ':') { err = "safetensors header: expected ':'"; return false; }
                ++j.p;
                bool ok = true;
                if (k == "dtype") { std::string d; ok = j.str(d); e.dtype = st_dtype(d); }
                else if (k == "shape") ok = j.int_array(e.shape);
                else if (k == "data_offsets") ok = j.int_array(offs);
                else ok = j.skip();
                if (!ok) { err = "safetensors header: " + j.err; return false; }
                j.ws();
                if (j.p < j.end && *j.p == ',') { ++j.p; continue; }
                if (j.p < j.end && *j.p == '}') { ++j.p; break; }
                err = "safetensors header: expected ',' or '}'"; return false;
            }
            const int es = elt_size(e.dtype);
            int64_t numel = 1;
            bool ovf = false;
            for (int64_t d : e.shape) {
                if (d < 0) { err = "negative dimension in '" + name + "'"; return false; }
                ovf |= __builtin_mul_overflow(numel, d, &numel);
            }
            int64_t nbytes = 0;
            ovf |= __builtin_mul_overflow(numel, (int64_t)es, &nbytes);
            if (ovf || es == 0 || offs.size() != 2 || offs[0] < 0 || offs[1] < offs[0] || (uint64_t)offs[1] > data_bytes || offs[1] - offs[0] != nbytes) {
                err = "tensor '" + name + "': dtype / shape / data_offsets are inconsistent"; return false;
            }
            e.stride.assign(e.shape.size(), 1);
            for (int d = (int)e.shape.size() - 2; d >= 0; --d) e.stride[d] = e.stride[d + 1] * e.shape[d + 1];
            e.data = base + offs[0];
            e.storage_bytes_left = offs[1] - offs[0];
            c.entries.push_back(std::move(e));
        }
        j.ws();
        if (j.p < j.end && *j.p == ',') { ++j.p; continue; }
        if (j.p < j.end && *j.p == '}') break;
        err = "safetensors header: expected ',' or '}'"; return false;
    }
    return true;
}

}  // namespace

extern "C" int mc_ckpt_open(const char* path, void** handle) {
    if (!path || !handle) { mc_set_error("mc_ckpt_open: null argument"); return 1; }
    auto c = std::make_unique<Ckpt>();
    c->fd = open(path, O_RDONLY);
    if (c->fd < 0) { mc_set_error("mc_ckpt_open: cannot open %s: %s", path, strerror(errno)); return 1; }
    struct stat st;
    if (fstat(c->fd, &st) != 0 || st.st_size <= 0) { mc_set_error("mc_ckpt_open: %s is empty or unreadable", path); return 1; }
    c->size = (size_t)st.st_size;
    // private copy-on-write mapping: pages come from the page cache, a stray write through a borrowed pointer cannot reach the file
    void* m = mmap(nullptr, c->size, PROT_READ | PROT_WRITE, MAP_PRIVATE, c->fd, 0);
    if (m == MAP_FAILED) { mc_set_error("mc_ckpt_open: mmap of %s failed: %s", path, strerror(errno)); return 1; }
    c->map = (uint8_t*)m;
    std::string err;
    const size_t len = strlen(path);
    const bool st_ext = len > 12 && strcmp(path + len - 12, ".safetensors") == 0;
    const bool is_zip = c->size >= 4 && rd32(c->map) == 0x04034b50u;
    // safetensors by CONTENT too (a file saved without the suffix): u64 header length inside the file, header opens a JSON object
    const bool st_sniff = !is_zip && c->size > 9 && rd64(c->map) <= (uint64_t)c->size - 8 && rd64(c->map) >= 2 && c->map[8] == '{';
    bool ok;
    if (is_zip) ok = load_torch_zip(*c, err);
    else if (st_ext || st_sniff) ok = load_safetensors(*c, err);
    else { ok = false; err = "neither a zip (torch.save) archive nor a .safetensors file"; }
    if (!ok) { mc_set_error("mc_ckpt_open: %s: %s", path, err.c_str()); return 1; }
    *handle = c.release();
    return 0;
}

extern "C" int mc_ckpt_close(void* handle) {
    delete (Ckpt*)handle;
    return 0;
}

extern "C" int mc_ckpt_count(void* handle, int* n) {
    if (!handle || !n) { mc_set_error("mc_ckpt_count: null argument"); return 1; }
    *n = (int)((Ckpt*)handle)->entries.size();
    return 0;
}

extern "C" int mc_ckpt_entry(void* handle, int index, const char** name, int* dtype, int* ndim, const int64_t** shape, const int64_t** strides,
                             const void** data, int64_t* storage_bytes) {
    Ckpt* c = (Ckpt*)handle;
    if (!c || index < 0 || index >= (int)c->entries.size()) { mc_set_error("mc_ckpt_entry: bad handle or index %d", index); return 1; }
    const Entry& e = c->entries[index];
    if (name) *name = e.name.c_str();
    if (dtype) *dtype = e.dtype;
    if (ndim) *ndim = (int)e.shape.size();
    if (shape) *shape = e.shape.data();
    if (strides) *strides = e.stride.data();
    if (data) *data = e.data;
    if (storage_bytes) *storage_bytes = e.storage_bytes_left;
    return 0;
}

extern "C" int mc_ckpt_entry_path(void* handle, int index, const char** path) {
    Ckpt* c = (Ckpt*)handle;
    if (!c || index < 0 || index >= (int)c->entries.size() || !path) { mc_set_error("mc_ckpt_entry_path: bad handle or index %d", index); return 1; }
    *path = c->entries[index].path.c_str();
    return 0;
}

extern "C" int mc_ckpt_scalar_count(void* handle, int* n) {
    if (!handle || !n) { mc_set_error("mc_ckpt_scalar_count: null argument"); return 1; }
    *n = (int)((Ckpt*)handle)->scalars.size();
    return 0;
}

extern "C" int mc_ckpt_scalar(void* handle, int index, const char** name, const char** path, int* kind, int64_t* ivalue, double* fvalue,
                              const char** svalue, int64_t* slen) {
    Ckpt* c = (Ckpt*)handle;
    if (!c || index < 0 || index >= (int)c->scalars.size()) { mc_set_error("mc_ckpt_scalar: bad handle or index %d", index); return 1; }
    const Scalar& sc = c->scalars[index];
    if (name) *name = sc.name.c_str();
    if (path) *path = sc.path.c_str();
    if (kind) *kind = sc.kind;
    if (ivalue) *ivalue = sc.i;
    if (fvalue) *fvalue = sc.f;
    if (svalue) *svalue = sc.s.data();
    if (slen) *slen = (int64_t)sc.s.size();
    return 0;
}

// Host-to-device copy of a CONTIGUOUS tensor straight from the mapped file (no intermediate host buffer); dst holds numel * element size bytes.
extern "C" int mc_ckpt_copy_to_device(void* handle, int index, void* dst_device, void* stream) {
    Ckpt* c = (Ckpt*)handle;
    if (!c || index < 0 || index >= (int)c->entries.size() || !dst_device) { mc_set_error("mc_ckpt_copy_to_device: bad arguments"); return 1; }
    const Entry& e = c->entries[index];
    int64_t numel = 1, expect = 1;
    for (int d = (int)e.shape.size() - 1; d >= 0; --d) {
        if (e.shape[d] != 1 && e.stride[d] != expect) { mc_set_error("mc_ckpt_copy_to_device: tensor '%s' is not contiguous", e.name.c_str()); return 1; }
        if (__builtin_mul_overflow(expect, e.shape[d], &expect) || __builtin_mul_overflow(numel, e.shape[d], &numel)) {
            mc_set_error("mc_ckpt_copy_to_device: tensor '%s' is too large", e.name.c_str()); return 1;
        }
    }
    if (numel == 0) return 0;
    int64_t nbytes;
    if (__builtin_mul_overflow(numel, (int64_t)elt_size(e.dtype), &nbytes) || nbytes > e.storage_bytes_left) {
        mc_set_error("mc_ckpt_copy_to_device: tensor '%s' reaches outside its storage record", e.name.c_str()); return 1;
    }
    hipError_t err = hipMemcpyAsync(dst_device, e.data, (size_t)nbytes, hipMemcpyHostToDevice, (hipStream_t)stream);
    if (err != hipSuccess) { mc_set_error("mc_ckpt_copy_to_device: %s", hipGetErrorString(err)); return 2; }
    return 0;
}
