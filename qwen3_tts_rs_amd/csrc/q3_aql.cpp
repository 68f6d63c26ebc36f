// q3_aql.cpp — the frame loop's own AQL submission path.
//
// A frame of generate_codes (lib.rs:580-652) is ~550 dependent kernel launches. hipGraphLaunch replays them as AQL dispatch
// packets whose headers HIP chooses: barrier bit set, agent-scope acquire and release fences on every packet. The fences are
// cache maintenance at every kernel boundary (L2 write-back of the producer XCD, L1 / scalar-cache / non-local L2 invalidation for
// the consumer) and cost ~0.3 us per dependent node on MI355X (tools/hw/aql_probe.hip, profiles/r4_aql_chain_probe.txt). A boundary
// can do without them when producer and consumer exchange their activations write-through (sc1 stores, drained) and L1-bypassing
// (sc1 loads): since round 6 the frame's GEMV and attention kernels do (q3_kernels.h "activation transport"), and this queue is the
// PRODUCT path of the frame loop — the engine's per-node policy (q3_session.hip: frame_fence_policy) drops the fences between
// those kernels and keeps them on every other node and on the first / last packet of a frame (state that crosses frames moves
// with plain accesses). B = 8 frame 2.71 -> 2.57 ms, codes identical. With HIP's own header policy (Q3_AQL=1) the
// path is bit-identical to and as fast as hipGraphLaunch — nothing in the frame time is runtime overhead —; packets without ANY
// fence (Q3_AQL=2) are a probe that gives wrong codes and needs the explicit opt-in Q3_AQL_UNSAFE=1.
// HIP offers no way to choose packet headers, so the engine converts its captured frame graph into a PACKET PROGRAM once
// and submits the packets itself:
//   * the kernels are the very same code objects: the .hip_fatbin section of this library is unbundled and loaded through the
//     HSA loader (hsa_executable_*), kernel descriptors are looked up by the names HIP reports for the graph's kernel nodes;
//   * kernel-argument blocks are packed from the nodes' parameter pointers using the argument table of the code object's
//     metadata note (offset / size / value_kind per argument, hidden arguments included), once, into device memory;
//   * one user-mode HSA queue per process and device carries the packets; a replay is a memcpy of the packet templates into the
//     ring plus one doorbell per frame; only the last packet of a submission carries a completion signal;
//   * per-node fence scopes come from the engine's policy (q3_aql.h), which knows which edges are write-through.
// libhsa-runtime64 is resolved with dlopen (it is the library libamdhip64 itself sits on, so it is always present and already
// initialised); libq3tts.so keeps linking only libamdhip64. Nothing here computes: without the HIP kernels there is no path.
#include "q3_aql.h"
#include "q3_capture_lock.h"

#include <dlfcn.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>
#include <link.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

namespace q3 {
namespace {

// ---- libhsa-runtime64 through dlopen -------------------------------------------------------------------------------------
struct Hsa {
    void* h = nullptr;
#define Q3_HSA_FN(name) decltype(&::name) name = nullptr;
    Q3_HSA_FN(hsa_init) Q3_HSA_FN(hsa_status_string) Q3_HSA_FN(hsa_iterate_agents) Q3_HSA_FN(hsa_agent_get_info)
    Q3_HSA_FN(hsa_code_object_reader_create_from_memory) Q3_HSA_FN(hsa_executable_create_alt) Q3_HSA_FN(hsa_executable_load_agent_code_object)
    Q3_HSA_FN(hsa_executable_freeze) Q3_HSA_FN(hsa_executable_get_symbol_by_name) Q3_HSA_FN(hsa_executable_symbol_get_info)
    Q3_HSA_FN(hsa_queue_create) Q3_HSA_FN(hsa_queue_destroy) Q3_HSA_FN(hsa_signal_create) Q3_HSA_FN(hsa_signal_destroy)
    Q3_HSA_FN(hsa_signal_store_relaxed) Q3_HSA_FN(hsa_signal_store_screlease) Q3_HSA_FN(hsa_signal_wait_scacquire) Q3_HSA_FN(hsa_signal_load_scacquire)
    Q3_HSA_FN(hsa_queue_load_read_index_scacquire) Q3_HSA_FN(hsa_queue_load_write_index_relaxed) Q3_HSA_FN(hsa_queue_store_write_index_relaxed)
#undef Q3_HSA_FN
    hsa_status_t (*hsa_amd_queue_cu_set_mask)(const hsa_queue_t*, uint32_t, const uint32_t*) = nullptr;      // optional (aql_restrict_cus)
    bool ok = false;
} hsa;

bool load_hsa(std::string* why) {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"libhsa-runtime64.so.1", "libhsa-runtime64.so"};
        for (const char* n : names) if (!hsa.h) hsa.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);     // the one libamdhip64 already carries
        for (const char* n : names) if (!hsa.h) hsa.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!hsa.h) return;
        bool all = true;
#define Q3_HSA_GET(name) hsa.name = (decltype(hsa.name))dlsym(hsa.h, #name); all = all && hsa.name;
        Q3_HSA_GET(hsa_init) Q3_HSA_GET(hsa_status_string) Q3_HSA_GET(hsa_iterate_agents) Q3_HSA_GET(hsa_agent_get_info)
        Q3_HSA_GET(hsa_code_object_reader_create_from_memory) Q3_HSA_GET(hsa_executable_create_alt) Q3_HSA_GET(hsa_executable_load_agent_code_object)
        Q3_HSA_GET(hsa_executable_freeze) Q3_HSA_GET(hsa_executable_get_symbol_by_name) Q3_HSA_GET(hsa_executable_symbol_get_info)
        Q3_HSA_GET(hsa_queue_create) Q3_HSA_GET(hsa_queue_destroy) Q3_HSA_GET(hsa_signal_create) Q3_HSA_GET(hsa_signal_destroy)
        Q3_HSA_GET(hsa_signal_store_relaxed) Q3_HSA_GET(hsa_signal_store_screlease) Q3_HSA_GET(hsa_signal_wait_scacquire) Q3_HSA_GET(hsa_signal_load_scacquire)
        Q3_HSA_GET(hsa_queue_load_read_index_scacquire) Q3_HSA_GET(hsa_queue_load_write_index_relaxed) Q3_HSA_GET(hsa_queue_store_write_index_relaxed)
#undef Q3_HSA_GET
        hsa.hsa_amd_queue_cu_set_mask = (decltype(hsa.hsa_amd_queue_cu_set_mask))dlsym(hsa.h, "hsa_amd_queue_cu_set_mask");
        hsa.ok = all && hsa.hsa_init() == HSA_STATUS_SUCCESS;       // reference-counted: HIP initialised ROCr first
    });
    if (!hsa.ok && why) *why = "libhsa-runtime64 is not available";
    return hsa.ok;
}
std::string hsa_err(const char* what, hsa_status_t st) {
    const char* m = nullptr;
    if (hsa.hsa_status_string) hsa.hsa_status_string(st, &m);
    return std::string(what) + ": " + (m ? m : "HSA error");
}

// ---- code-object metadata (the msgpack note NT_AMDGPU_METADATA): name -> argument table -----------------------------------
struct ArgInfo { uint32_t offset = 0, size = 0; std::string kind; };
struct KernelMeta { std::string name, symbol; uint32_t kernarg_size = 0; std::vector<ArgInfo> args; };

struct Mp {                                                  // the subset of msgpack the metadata note uses
    const uint8_t* p; const uint8_t* e; bool bad = false;
    uint64_t be(int n) { uint64_t v = 0; if (p + n > e) { bad = true; return 0; } for (int i = 0; i < n; ++i) v = (v << 8) | *p++; return v; }
    uint8_t peek() { if (p >= e) { bad = true; return 0xc0; } return *p; }
    bool is_map() { const uint8_t t = peek(); return (t & 0xf0) == 0x80 || t == 0xde || t == 0xdf; }
    bool is_arr() { const uint8_t t = peek(); return (t & 0xf0) == 0x90 || t == 0xdc || t == 0xdd; }
    bool is_str() { const uint8_t t = peek(); return (t & 0xe0) == 0xa0 || t == 0xd9 || t == 0xda || t == 0xdb; }
    size_t map_n() { const uint8_t t = (uint8_t)be(1); return (t & 0xf0) == 0x80 ? (t & 15) : t == 0xde ? be(2) : be(4); }
    size_t arr_n() { const uint8_t t = (uint8_t)be(1); return (t & 0xf0) == 0x90 ? (t & 15) : t == 0xdc ? be(2) : be(4); }
    std::string str() {
        const uint8_t t = (uint8_t)be(1);
        const size_t n = (t & 0xe0) == 0xa0 ? (t & 31) : t == 0xd9 ? be(1) : t == 0xda ? be(2) : be(4);
        if (p + n > e) { bad = true; return {}; }
        std::string s((const char*)p, n); p += n; return s;
    }
    bool is_int() { const uint8_t t = peek(); return t < 0x80 || t >= 0xe0 || (t >= 0xcc && t <= 0xd3); }
    int64_t integer() {
        const uint8_t t = (uint8_t)be(1);
        if (t < 0x80) return t;
        if (t >= 0xe0) return (int8_t)t;
        switch (t) {
            case 0xcc: return (int64_t)be(1); case 0xcd: return (int64_t)be(2); case 0xce: return (int64_t)be(4); case 0xcf: return (int64_t)be(8);
            case 0xd0: return (int8_t)be(1); case 0xd1: return (int16_t)be(2); case 0xd2: return (int32_t)be(4); case 0xd3: return (int64_t)be(8);
        }
        bad = true; return 0;
    }
    void skip() {
        if (bad) return;
        if (is_map()) { size_t n = map_n(); for (size_t i = 0; i < n && !bad; ++i) { skip(); skip(); } return; }
        if (is_arr()) { size_t n = arr_n(); for (size_t i = 0; i < n && !bad; ++i) skip(); return; }
        if (is_str()) { (void)str(); return; }
        if (is_int()) { (void)integer(); return; }
        const uint8_t t = (uint8_t)be(1);
        switch (t) {
            case 0xc0: case 0xc2: case 0xc3: return;
            case 0xca: p += 4; return; case 0xcb: p += 8; return;
            case 0xc4: { size_t n = be(1); p += n; return; } case 0xc5: { size_t n = be(2); p += n; return; } case 0xc6: { size_t n = be(4); p += n; return; }
            default: bad = true;
        }
    }
};

bool parse_metadata(const uint8_t* elf, size_t size, std::vector<KernelMeta>* out) {
    if (size < 64 || memcmp(elf, "\177ELF", 4) != 0 || elf[4] != 2) return false;
    uint64_t shoff; uint16_t shentsize, shnum;
    memcpy(&shoff, elf + 0x28, 8); memcpy(&shentsize, elf + 0x3a, 2); memcpy(&shnum, elf + 0x3c, 2);
    for (unsigned i = 0; i < shnum; ++i) {
        const uint8_t* sh = elf + shoff + (size_t)i * shentsize;
        if (sh + 64 > elf + size) return false;
        uint32_t type; uint64_t off, sz;
        memcpy(&type, sh + 4, 4); memcpy(&off, sh + 0x18, 8); memcpy(&sz, sh + 0x20, 8);
        if (type != 7 /* SHT_NOTE */ || off + sz > size) continue;
        const uint8_t* n = elf + off; const uint8_t* ne = n + sz;
        while (n + 12 <= ne) {
            uint32_t namesz, descsz, ntype;
            memcpy(&namesz, n, 4); memcpy(&descsz, n + 4, 4); memcpy(&ntype, n + 8, 4);
            const uint8_t* name = n + 12; const uint8_t* desc = name + ((namesz + 3) & ~3u);
            if (desc + descsz > ne) break;
            if (ntype == 32 && namesz >= 6 && memcmp(name, "AMDGPU", 6) == 0) {
                Mp m{desc, desc + descsz};
                if (!m.is_map()) return false;
                const size_t top = m.map_n();
                for (size_t t = 0; t < top && !m.bad; ++t) {
                    const std::string key = m.str();
                    if (key != "amdhsa.kernels" || !m.is_arr()) { m.skip(); continue; }
                    const size_t nk = m.arr_n();
                    for (size_t k = 0; k < nk && !m.bad; ++k) {
                        KernelMeta km;
                        const size_t nf = m.map_n();
                        for (size_t f = 0; f < nf && !m.bad; ++f) {
                            const std::string fk = m.str();
                            if (fk == ".name") km.name = m.str();
                            else if (fk == ".symbol") km.symbol = m.str();
                            else if (fk == ".kernarg_segment_size") km.kernarg_size = (uint32_t)m.integer();
                            else if (fk == ".args" && m.is_arr()) {
                                const size_t na = m.arr_n();
                                for (size_t a = 0; a < na && !m.bad; ++a) {
                                    ArgInfo ai;
                                    const size_t naf = m.map_n();
                                    for (size_t q = 0; q < naf && !m.bad; ++q) {
                                        const std::string ak = m.str();
                                        if (ak == ".offset") ai.offset = (uint32_t)m.integer();
                                        else if (ak == ".size") ai.size = (uint32_t)m.integer();
                                        else if (ak == ".value_kind") ai.kind = m.str();
                                        else m.skip();
                                    }
                                    km.args.push_back(ai);
                                }
                            } else m.skip();
                        }
                        out->push_back(std::move(km));
                    }
                }
                return !m.bad;
            }
            n = desc + ((descsz + 3) & ~3u);
        }
    }
    return false;
}

// ---- this library's own device code: the clang offload bundles of .hip_fatbin ---------------------------------------------
bool own_code_objects(std::vector<std::vector<uint8_t>>* out, std::string* why) {
    Dl_info di;
    if (!dladdr((void*)&own_code_objects, &di) || !di.dli_fname) { *why = "dladdr failed"; return false; }
    FILE* f = fopen(di.dli_fname, "rb");
    if (!f) { *why = std::string("cannot read ") + di.dli_fname; return false; }
    fseek(f, 0, SEEK_END); const long sz = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<uint8_t> file((size_t)sz);
    const bool rd = fread(file.data(), 1, (size_t)sz, f) == (size_t)sz; fclose(f);
    if (!rd || sz < 64 || memcmp(file.data(), "\177ELF", 4) != 0) { *why = "not an ELF file"; return false; }
    uint64_t shoff; uint16_t shentsize, shnum, shstrndx;
    memcpy(&shoff, &file[0x28], 8); memcpy(&shentsize, &file[0x3a], 2); memcpy(&shnum, &file[0x3c], 2); memcpy(&shstrndx, &file[0x3e], 2);
    auto sh = [&](unsigned i) { return file.data() + shoff + (size_t)i * shentsize; };
    uint64_t stroff; memcpy(&stroff, sh(shstrndx) + 0x18, 8);
    const uint8_t* fat = nullptr; uint64_t fat_size = 0;
    for (unsigned i = 0; i < shnum; ++i) {
        uint32_t nm; uint64_t off, ssz;
        memcpy(&nm, sh(i), 4); memcpy(&off, sh(i) + 0x18, 8); memcpy(&ssz, sh(i) + 0x20, 8);
        if (strcmp((const char*)file.data() + stroff + nm, ".hip_fatbin") == 0) { fat = file.data() + off; fat_size = ssz; }
    }
    if (!fat) { *why = "no .hip_fatbin section"; return false; }
    static const char magic[] = "__CLANG_OFFLOAD_BUNDLE__";
    for (uint64_t at = 0; at + 32 <= fat_size;) {
        if (memcmp(fat + at, magic, 24) != 0) { at += 8; continue; }
        const uint8_t* b = fat + at; uint64_t n; memcpy(&n, b + 24, 8);
        const uint8_t* e = b + 32; uint64_t end = 32;
        for (uint64_t i = 0; i < n; ++i) {
            uint64_t off, size, tl; memcpy(&off, e, 8); memcpy(&size, e + 8, 8); memcpy(&tl, e + 16, 8);
            const std::string triple((const char*)e + 24, (size_t)tl);
            e += 24 + tl;
            if (triple.find("amdgcn-amd-amdhsa") != std::string::npos && triple.find("gfx950") != std::string::npos && size)
                out->emplace_back(b + off, b + off + size);
            if (off + size > end) end = off + size;
        }
        at += (end + 7) & ~(uint64_t)7;
    }
    if (out->empty()) { *why = "no gfx950 code object in .hip_fatbin"; return false; }
    return true;
}

// ---- per-device runtime: agent, executable, queue --------------------------------------------------------------------------
struct KernelSym { uint64_t object = 0; uint32_t kernarg = 0, group = 0, priv = 0; const KernelMeta* meta = nullptr; };
struct Runtime {
    int device = -1; hsa_agent_t agent{}; hsa_executable_t exe{};
    std::vector<std::vector<uint8_t>> blobs; std::vector<KernelMeta> metas;
    std::map<std::string, KernelSym> kernels;            // by the name HIP reports
    hsa_queue_t* queue = nullptr; uint64_t write_idx = 0;
    std::mutex mu;                                       // one submitter at a time
    int restricted = 0;                                  // aql_restrict_cus holders (the mask is the queue's: every program on it sees it)
};
std::mutex g_mu;
std::map<int, Runtime*> g_rt;

struct FindAgent { uint32_t bdf; uint32_t domain; hsa_agent_t agent; bool found; };
hsa_status_t agent_cb(hsa_agent_t a, void* data) {
    auto* fa = (FindAgent*)data;
    hsa_device_type_t t;
    if (hsa.hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS || t != HSA_DEVICE_TYPE_GPU) return HSA_STATUS_SUCCESS;
    uint32_t bdf = 0, dom = 0;
    hsa.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf);
    hsa.hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &dom);
    if (bdf == fa->bdf && dom == fa->domain && !fa->found) { fa->agent = a; fa->found = true; }
    return HSA_STATUS_SUCCESS;
}

Runtime* runtime_for(int device, std::string* why) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_rt.find(device);
    if (it != g_rt.end()) return it->second;
    if (!load_hsa(why)) return nullptr;
    char bus[64] = {0};
    if (hipDeviceGetPCIBusId(bus, sizeof bus, device) != hipSuccess) { *why = "hipDeviceGetPCIBusId failed"; return nullptr; }
    unsigned dom = 0, b = 0, d = 0, fn = 0;
    if (sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &fn) != 4) { *why = std::string("unparsable PCI bus id ") + bus; return nullptr; }
    FindAgent fa{(b << 8) | (d << 3) | fn, dom, {}, false};
    hsa.hsa_iterate_agents(agent_cb, &fa);
    if (!fa.found) { *why = std::string("no HSA GPU agent at ") + bus; return nullptr; }
    std::unique_ptr<Runtime> rt(new Runtime());
    rt->device = device; rt->agent = fa.agent;
    if (!own_code_objects(&rt->blobs, why)) return nullptr;
    hsa_status_t st = hsa.hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &rt->exe);
    if (st != HSA_STATUS_SUCCESS) { *why = hsa_err("hsa_executable_create_alt", st); return nullptr; }
    for (auto& blob : rt->blobs) {
        hsa_code_object_reader_t rd;
        st = hsa.hsa_code_object_reader_create_from_memory(blob.data(), blob.size(), &rd);
        if (st != HSA_STATUS_SUCCESS) { *why = hsa_err("hsa_code_object_reader_create_from_memory", st); return nullptr; }
        st = hsa.hsa_executable_load_agent_code_object(rt->exe, rt->agent, rd, nullptr, nullptr);
        if (st != HSA_STATUS_SUCCESS) { *why = hsa_err("hsa_executable_load_agent_code_object", st); return nullptr; }
        if (!parse_metadata(blob.data(), blob.size(), &rt->metas)) { *why = "code-object metadata note not understood"; return nullptr; }
    }
    st = hsa.hsa_executable_freeze(rt->exe, nullptr);
    if (st != HSA_STATUS_SUCCESS) { *why = hsa_err("hsa_executable_freeze", st); return nullptr; }
    for (const KernelMeta& km : rt->metas) {
        hsa_executable_symbol_t sym;
        if (hsa.hsa_executable_get_symbol_by_name(rt->exe, km.symbol.c_str(), &rt->agent, &sym) != HSA_STATUS_SUCCESS) continue;
        KernelSym ks; ks.meta = &km;
        hsa.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &ks.object);
        hsa.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &ks.kernarg);
        hsa.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &ks.group);
        hsa.hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &ks.priv);
        rt->kernels[km.name] = ks;
    }
    uint32_t qmax = 0;
    hsa.hsa_agent_get_info(rt->agent, HSA_AGENT_INFO_QUEUE_MAX_SIZE, &qmax);
    uint32_t qsize = 65536; while (qsize > qmax && qsize > 1024) qsize >>= 1;
    st = hsa.hsa_queue_create(rt->agent, qsize, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &rt->queue);
    if (st != HSA_STATUS_SUCCESS) { *why = hsa_err("hsa_queue_create", st); return nullptr; }
    Runtime* raw = rt.release();
    g_rt[device] = raw;                                  // lives until the process ends
    return raw;
}

}  // namespace

struct AqlProgram {
    Runtime* rt = nullptr;
    std::vector<hsa_kernel_dispatch_packet_t> pk;        // templates (header / setup filled, completion signal empty)
    void* kernargs = nullptr; bool host_kernargs = false;     // one block per node: device memory (pinned host memory with Q3_AQL_HOST_KERNARG=1)
    hsa_signal_t done{}; bool pending = false;
    int n_acq_free = 0, n_rel_free = 0;                  // nodes submitted without their acquire / release fence (node_policy)
    bool dead = false;      // a submission timed out: the ring may hold a partial burst without a completion signal — never submit or wait again
};

int aql_program_nodes(const AqlProgram* p) { return p ? (int)p->pk.size() : 0; }
void aql_program_fence_free(const AqlProgram* p, int* acquire_free, int* release_free) {
    if (acquire_free) *acquire_free = p ? p->n_acq_free : 0;
    if (release_free) *release_free = p ? p->n_rel_free : 0;
}

void aql_program_destroy(AqlProgram* p) {
    if (!p) return;
    std::string w;
    if (p->pending && !p->dead) aql_wait(p, &w);         // (a dead program's signal may never fire: do not block on it)
    if (p->kernargs && !p->dead) (void)(p->host_kernargs ? hipHostFree(p->kernargs) : hipFree(p->kernargs));      // dead: packets still in the ring may point at the blocks — leak them
    if (p->dead) p->kernargs = nullptr;
    if (p->done.handle && !p->dead) hsa.hsa_signal_destroy(p->done);     // dead: the burst's last packet may still name it — leaked with the blocks
    delete p;
}

AqlProgram* aql_program_create(hipGraph_t graph, int device, const AqlPolicy& pol, std::string* why) {
    std::string dummy; if (!why) why = &dummy;
    Runtime* rt = runtime_for(device, why);
    if (!rt) return nullptr;
    size_t n = 0;
    if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess || n == 0) { *why = "hipGraphGetNodes failed"; return nullptr; }
    std::vector<hipGraphNode_t> nodes(n);
    if (hipGraphGetNodes(graph, nodes.data(), &n) != hipSuccess) { *why = "hipGraphGetNodes failed"; return nullptr; }
    // topological order (the captured frame is a chain; every packet carries the barrier bit, so any topological order is valid)
    std::map<hipGraphNode_t, size_t> index; for (size_t i = 0; i < n; ++i) index[nodes[i]] = i;
    std::vector<std::vector<size_t>> deps(n);
    for (size_t i = 0; i < n; ++i) {
        size_t nd = 0;
        if (hipGraphNodeGetDependencies(nodes[i], nullptr, &nd) != hipSuccess) { *why = "hipGraphNodeGetDependencies failed"; return nullptr; }
        std::vector<hipGraphNode_t> dn(nd);
        if (nd && hipGraphNodeGetDependencies(nodes[i], dn.data(), &nd) != hipSuccess) { *why = "hipGraphNodeGetDependencies failed"; return nullptr; }
        for (auto d : dn) { auto f = index.find(d); if (f == index.end()) { *why = "dependency outside the graph"; return nullptr; } deps[i].push_back(f->second); }
    }
    std::vector<size_t> order; std::vector<char> placed(n, 0);
    while (order.size() < n) {
        bool any = false;
        for (size_t i = 0; i < n; ++i) {
            if (placed[i]) continue;
            bool ready = true; for (size_t d : deps[i]) ready = ready && placed[d];
            if (ready) { placed[i] = 1; order.push_back(i); any = true; }
        }
        if (!any) { *why = "graph has a cycle"; return nullptr; }
    }
    std::unique_ptr<AqlProgram> p(new AqlProgram());
    p->rt = rt;
    std::vector<uint8_t> host_args; std::vector<size_t> arg_off;
    for (size_t oi : order) {
        hipGraphNodeType ty;
        if (hipGraphNodeGetType(nodes[oi], &ty) != hipSuccess) { *why = "hipGraphNodeGetType failed"; return nullptr; }
        if (ty == hipGraphNodeTypeEmpty) continue;
        if (ty != hipGraphNodeTypeKernel) { *why = "graph node of type " + std::to_string((int)ty) + " (only kernel nodes convert)"; return nullptr; }
        hipKernelNodeParams kp{};
        if (hipGraphKernelNodeGetParams(nodes[oi], &kp) != hipSuccess) { *why = "hipGraphKernelNodeGetParams failed"; return nullptr; }
        if (kp.extra || !kp.kernelParams) { *why = "kernel node without a parameter array"; return nullptr; }
        const char* nm = hipKernelNameRefByPtr(kp.func, nullptr);
        if (!nm) { *why = "hipKernelNameRefByPtr returned no name"; return nullptr; }
        auto kit = rt->kernels.find(nm);
        if (kit == rt->kernels.end()) { *why = std::string("kernel not found in the loaded code objects: ") + nm; return nullptr; }
        const KernelSym& ks = kit->second;
        const size_t base = (host_args.size() + 255) & ~(size_t)255;
        host_args.resize(base + ((ks.kernarg + 255) & ~255u), 0);
        uint8_t* ka = host_args.data() + base;
        size_t explicit_i = 0;
        for (const ArgInfo& ai : ks.meta->args) {
            if (ai.offset + ai.size > ks.kernarg) { *why = "argument beyond the kernarg segment"; return nullptr; }
            auto put = [&](uint64_t v) { memcpy(ka + ai.offset, &v, ai.size < 8 ? ai.size : 8); };
            const std::string& k = ai.kind;
            if (k.compare(0, 7, "hidden_") != 0) { memcpy(ka + ai.offset, kp.kernelParams[explicit_i++], ai.size); continue; }
            if (k == "hidden_block_count_x") put(kp.gridDim.x); else if (k == "hidden_block_count_y") put(kp.gridDim.y); else if (k == "hidden_block_count_z") put(kp.gridDim.z);
            else if (k == "hidden_group_size_x") put(kp.blockDim.x); else if (k == "hidden_group_size_y") put(kp.blockDim.y); else if (k == "hidden_group_size_z") put(kp.blockDim.z);
            else if (k == "hidden_remainder_x" || k == "hidden_remainder_y" || k == "hidden_remainder_z") put(0);
            else if (k == "hidden_global_offset_x" || k == "hidden_global_offset_y" || k == "hidden_global_offset_z") put(0);
            else if (k == "hidden_grid_dims") put(kp.gridDim.z > 1 ? 3 : kp.gridDim.y > 1 ? 2 : 1);
            else if (k == "hidden_dynamic_lds_size") put(kp.sharedMemBytes);
            else if (k == "hidden_none") { }
            else { *why = std::string(nm) + " uses " + k + " (not provided by this submission path)"; return nullptr; }
        }
        hsa_kernel_dispatch_packet_t pk; memset(&pk, 0, sizeof pk);
        int want_acq = pol.acquire, want_rel = pol.release;
        if (pol.node_policy) pol.node_policy(nm, &want_acq, &want_rel);
        const int acq = want_acq ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE, rel = want_rel ? HSA_FENCE_SCOPE_AGENT : HSA_FENCE_SCOPE_NONE;
        p->n_acq_free += want_acq ? 0 : 1; p->n_rel_free += want_rel ? 0 : 1;
        pk.header = (uint16_t)((HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                               (acq << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) | (rel << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE));
        pk.setup = 3 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        pk.workgroup_size_x = (uint16_t)kp.blockDim.x; pk.workgroup_size_y = (uint16_t)kp.blockDim.y; pk.workgroup_size_z = (uint16_t)kp.blockDim.z;
        pk.grid_size_x = kp.gridDim.x * kp.blockDim.x; pk.grid_size_y = kp.gridDim.y * kp.blockDim.y; pk.grid_size_z = kp.gridDim.z * kp.blockDim.z;
        pk.private_segment_size = ks.priv; pk.group_segment_size = ks.group + kp.sharedMemBytes;
        pk.kernel_object = ks.object;
        p->pk.push_back(pk); arg_off.push_back(base);
    }
    if (p->pk.empty()) { *why = "graph holds no kernel node"; return nullptr; }
    if (pol.node_policy) {
        // frame boundary: whatever crosses frames (counters, positions, the talker's K/V, codes) moves with plain accesses
        auto set = [](uint16_t& h, int shift) { h = (uint16_t)((h & ~(3u << shift)) | ((unsigned)HSA_FENCE_SCOPE_AGENT << shift)); };
        set(p->pk.front().header, HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE);
        set(p->pk.back().header, HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    }
    // Kernel-argument blocks live in DEVICE memory: the dispatcher reads a node's first 14 dwords into SGPRs before its first
    // wave starts, and that fetch sits in every node boundary — with the blocks in pinned host memory (Q3_AQL_HOST_KERNARG=1: an
    // A/B aid) the B = 8 frame is 14 % slower (2.567 -> 2.922 ms, profiles/r6_kernarg_placement_ab.txt).
    p->host_kernargs = getenv("Q3_AQL_HOST_KERNARG") && atoi(getenv("Q3_AQL_HOST_KERNARG")) == 1;
    if (p->host_kernargs) {
        if (hipHostMalloc(&p->kernargs, host_args.size(), hipHostMallocDefault) != hipSuccess) { *why = "hipHostMalloc(kernargs) failed"; return nullptr; }
        memcpy(p->kernargs, host_args.data(), host_args.size());
    } else {
        if (hipMalloc(&p->kernargs, host_args.size()) != hipSuccess) { *why = "hipMalloc(kernargs) failed"; return nullptr; }
        if (q3_hipMemcpy(p->kernargs, host_args.data(), host_args.size(), hipMemcpyHostToDevice) != hipSuccess) { *why = "q3_hipMemcpy(kernargs) failed"; aql_program_destroy(p.release()); return nullptr; }
    }
    for (size_t i = 0; i < p->pk.size(); ++i) p->pk[i].kernarg_address = (char*)p->kernargs + arg_off[i];
    if ((size_t)p->pk.size() * 2 > rt->queue->size) { *why = "frame longer than half the packet ring"; aql_program_destroy(p.release()); return nullptr; }
    const hsa_status_t st = hsa.hsa_signal_create(0, 0, nullptr, &p->done);
    if (st != HSA_STATUS_SUCCESS) { *why = hsa_err("hsa_signal_create", st); aql_program_destroy(p.release()); return nullptr; }
    return p.release();
}

bool aql_submit(AqlProgram* p, int frames, std::string* why, int* submitted) {
    std::string dummy; if (!why) why = &dummy;
    if (submitted) *submitted = 0;
    if (!p || frames <= 0) { *why = "aql_submit: bad argument"; return false; }
    if (p->dead) { *why = "aql_submit: this program's queue stopped draining earlier"; return false; }
    if (p->pending && !aql_wait(p, why)) return false;
    Runtime* rt = p->rt;
    hsa_queue_t* q = rt->queue;
    const uint32_t mask = q->size - 1; const size_t n = p->pk.size();
    auto* ring = (hsa_kernel_dispatch_packet_t*)q->base_address;
    hsa.hsa_signal_store_relaxed(p->done, 1);
    p->pending = true;
    auto t0 = std::chrono::steady_clock::now();                 // of the last sign of life of the ring (the read index moved)
    uint64_t last_rd = hsa.hsa_queue_load_read_index_scacquire(q);
    for (int f = 0; f < frames; ++f) {
        // The queue is shared by every session of the process on this device: the lock is taken PER FRAME, so bursts of sessions
        // driven from different host threads interleave at frame boundaries (a 640-frame burst used to hold it for its whole
        // 1.6 s), and it is never held while waiting for ring space. A frame's packets are contiguous in the ring and its first /
        // last packet keep their fences, so another session's frame in between changes nothing for this one.
        for (;;) {
            std::unique_lock<std::mutex> lk(rt->mu);
            if (rt->write_idx + n - hsa.hsa_queue_load_read_index_scacquire(q) <= q->size) {      // room for one frame (the ring holds at least two)
                for (size_t i = 0; i < n; ++i) {
                    hsa_kernel_dispatch_packet_t pk = p->pk[i];
                    const bool first = f == 0 && i == 0, last = f == frames - 1 && i == n - 1;
                    uint16_t hdr = pk.header;
                    if (first) hdr = (uint16_t)((hdr & ~(3u << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE)) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE));
                    if (last) { hdr = (uint16_t)((hdr & ~(3u << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE)) | (HSA_FENCE_SCOPE_SYSTEM << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE)); pk.completion_signal = p->done; }
                    hsa_kernel_dispatch_packet_t* dst = ring + ((rt->write_idx + i) & mask);
                    memcpy((char*)dst + 4, (char*)&pk + 4, sizeof pk - 4);
                    __atomic_store_n(&dst->full_header, (uint32_t)hdr | ((uint32_t)pk.setup << 16), __ATOMIC_RELEASE);
                }
                rt->write_idx += n;
                hsa.hsa_queue_store_write_index_relaxed(q, rt->write_idx);
                hsa.hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)(rt->write_idx - 1));
                break;
            }
            lk.unlock();
            {   // a long burst waits here most of its time (the ring holds ~119 frames): only a ring that has not moved for 120 s is dead
                const uint64_t rd = hsa.hsa_queue_load_read_index_scacquire(q);
                if (rd != last_rd) { last_rd = rd; t0 = std::chrono::steady_clock::now(); }
            }
            if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) {
                // f whole frames are in the ring and none of them carries the completion signal (only the burst's last packet
                // would have): the program is dead — nothing waits on its signal any more, the caller accounts for the f frames
                // that were handed to the device and leaves this path
                *why = "aql_submit: the packet ring did not drain within 120 s";
                p->dead = true; p->pending = false;
                if (submitted) *submitted = f;
                return false;
            }
            std::this_thread::yield();
        }
    }
    if (submitted) *submitted = frames;
    return true;
}

// The CU mask is a property of the hardware queue (KFD rewrites the queue's descriptor: an ioctl, ~0.1 ms — not for the inner loop).
// Bit i names CU i / 8 of XCD i % 8 on this part (KFD deals the bits round the XCDs, and inside one round its shader engines), so the
// first n bits are n / 8 CUs of EVERY XCD and workgroup b still lands on XCD b % 8.
bool aql_restrict_cus(AqlProgram* p, int first_cus, std::string* why) {
    std::string dummy; if (!why) why = &dummy;
    if (!p || !p->rt) { *why = "aql_restrict_cus: no program"; return false; }
    if (!hsa.hsa_amd_queue_cu_set_mask) { *why = "hsa_amd_queue_cu_set_mask is not exported by this ROCr"; return false; }
    Runtime* rt = p->rt;
    std::lock_guard<std::mutex> lk(rt->mu);
    const bool release = first_cus <= 0;
    if (release) { if (rt->restricted == 0 || --rt->restricted > 0) return true; }
    else if (rt->restricted++ > 0) return true;          // the first holder's mask stays until the last one lets go
    uint32_t mask[8];
    for (int w = 0; w < 8; ++w) {
        const int lo = w * 32, n = release ? 32 : (first_cus - lo < 0 ? 0 : (first_cus - lo > 32 ? 32 : first_cus - lo));
        mask[w] = n >= 32 ? 0xffffffffu : ((1u << n) - 1u);
    }
    const hsa_status_t st = hsa.hsa_amd_queue_cu_set_mask(rt->queue, 256, mask);
    if (st != HSA_STATUS_SUCCESS) { if (!release) --rt->restricted; *why = hsa_err("hsa_amd_queue_cu_set_mask", st); return false; }
    return true;
}

bool aql_wait(AqlProgram* p, std::string* why) {
    std::string dummy; if (!why) why = &dummy;
    if (!p || !p->pending) return true;
    if (p->dead) { *why = "aql_wait: dead program"; return false; }
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hsa_signal_value_t v = hsa.hsa_signal_wait_scacquire(p->done, HSA_SIGNAL_CONDITION_LT, 1, 2000000000ull, HSA_WAIT_STATE_BLOCKED);
        if (v < 1) break;
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(300)) { *why = "aql_wait: no completion within 300 s"; p->dead = true; p->pending = false; return false; }
    }
    p->pending = false;
    return true;
}

}  // namespace q3
