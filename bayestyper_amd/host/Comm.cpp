#include "Comm.hpp"
#include "Parallel.hpp"

#include <dirent.h>
#include <iterator>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cerrno>

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <numeric>
#include <stdexcept>
#include <thread>

#include "../../include/btcomm.h"
#include "InferenceEngine.hpp"
#include "KmerCounter.hpp"

namespace bthost {

struct Comm::Api {
    int (*unique_id)(uint8_t *);
    int (*init)(bt_ctx *, const uint8_t *, int, int, bt_comm **);
    int (*destroy)(bt_comm *);
    int (*allreduce_hist)(bt_comm *, uint64_t *, uint64_t);
    int (*gather_summaries)(bt_comm *, const uint32_t *, uint64_t, uint32_t *, uint64_t, uint64_t *);
    int (*allgatherv)(bt_comm *, const uint8_t *, uint64_t, uint8_t *, uint64_t, uint64_t *);
};

namespace {
void check(int rc, const char *what) {
    if (rc != BT_OK) throw std::runtime_error(std::string(what) + ": " + bt_last_error());
}
int envInt(const char *name, int dflt) {
    const char *e = getenv(name);
    return e && *e ? atoi(e) : dflt;
}
std::string libraryDir() {   // the directory of libbthost.so (libbtcomm.so sits next to it)
    Dl_info info;
    if (dladdr((const void *)&envInt, &info) && info.dli_fname) {
        std::string p = info.dli_fname;
        const size_t slash = p.rfind('/');
        if (slash != std::string::npos) return p.substr(0, slash);
    }
    return ".";
}
struct DeviceBuffer {
    bt_ctx *ctx;
    void *p = nullptr;
    DeviceBuffer(bt_ctx *c, size_t bytes) : ctx(c) { check(bt_malloc(ctx, std::max<size_t>(bytes, 16), &p), "bt_malloc"); }
    ~DeviceBuffer() { bt_free(ctx, p); }
};
}  // namespace

int Comm::envRank() { return envInt("BT_RANK", 0); }
void Comm::markFailed() {
    const char *id_file = getenv("BT_COMM_ID_FILE");
    if (envWorld() <= 1 || !id_file || !*id_file) return;
    std::ofstream f(std::string(id_file) + ".failed");
    f << "rank " << envRank() << "\n";
}
int Comm::envWorld() { return std::max(1, envInt("BT_WORLD", 1)); }

std::unique_ptr<Comm> Comm::fromEnvironment(bt_ctx *ctx) {
    const int world = envWorld(), rank = envRank();
    if (world <= 1) return nullptr;
    if (rank < 0 || rank >= world) throw std::runtime_error("BT_RANK must be in [0, BT_WORLD)");
    const char *id_file = getenv("BT_COMM_ID_FILE");
    if (!id_file || !*id_file) throw std::runtime_error("BT_WORLD > 1 needs BT_COMM_ID_FILE (a path all ranks can read; rank 0 writes the communicator id there)");
    std::unique_ptr<Comm> c(new Comm());
    c->ctx = ctx;
    c->rank_ = rank;
    c->world_ = world;
    c->id_file_ = id_file;
    // A run's rendezvous carries the run's nonce (BT_COMM_NONCE, the same on every rank: the starting process sets it; an external launcher should): what a
    // previous run left under the same path — the old communicator id, a "failed" marker — is then never taken for this run's.  Without a nonce rank 0 still
    // removes the leftovers before it publishes, which covers every order of events except a rank > 0 reading the stale id before rank 0 got that far.
    const std::string nonce = getenv("BT_COMM_NONCE") ? getenv("BT_COMM_NONCE") : "";
    if (rank == 0) {
        std::remove(id_file);
        std::remove((std::string(id_file) + ".failed").c_str());
    }
    const char *transport = getenv("BT_COMM_TRANSPORT");
    if (transport && std::strcmp(transport, "files") == 0) {   // tests: ranks sharing one GPU
        c->dir = std::string(id_file) + ".d";
        if (rank == 0) {
            if (mkdir(c->dir.c_str(), 0777) != 0 && errno != EEXIST) throw std::runtime_error("cannot create " + c->dir);
            // (leftovers of an earlier run under this path: its exchange files would be taken for this run's)
            if (DIR *dh = opendir(c->dir.c_str())) {
                while (dirent *e = readdir(dh))
                    if (e->d_name[0] != '.') std::remove((c->dir + "/" + e->d_name).c_str());
                closedir(dh);
            }
            const std::string tmp = std::string(id_file) + ".tmp";
            {
                std::ofstream f(tmp);   // the other ranks wait for this file like for the communicator id
                f << "files " << nonce << "\n";
            }
            if (std::rename(tmp.c_str(), id_file) != 0) throw std::runtime_error(std::string("cannot write ") + id_file);
        } else {
            bool got = false;
            for (int tries = 0; tries < 6000 && !got; ++tries) {
                std::ifstream f(id_file);
                std::string word, n;
                if (f && (f >> word) && word == "files") {
                    f >> n;
                    got = n == nonce;
                }
                if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            if (!got) throw std::runtime_error(std::string("rank ") + std::to_string(rank) + ": " + id_file + " did not appear");
        }
        return c;
    }
    if (transport && *transport && std::strcmp(transport, "rccl") != 0) throw std::runtime_error("BT_COMM_TRANSPORT must be rccl (default) or files");
    const std::string path = libraryDir() + "/libbtcomm.so";
    c->dl = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!c->dl) throw std::runtime_error(std::string("cannot load ") + path + ": " + dlerror());
    c->api.reset(new Api());
    auto sym = [&](const char *name) {
        void *p = dlsym(c->dl, name);
        if (!p) throw std::runtime_error(std::string("libbtcomm.so lacks ") + name);
        return p;
    };
    c->api->unique_id = (int (*)(uint8_t *))sym("bt_comm_unique_id");
    c->api->init = (int (*)(bt_ctx *, const uint8_t *, int, int, bt_comm **))sym("bt_comm_init");
    c->api->destroy = (int (*)(bt_comm *))sym("bt_comm_destroy");
    c->api->allreduce_hist = (int (*)(bt_comm *, uint64_t *, uint64_t))sym("bt_comm_allreduce_hist");
    c->api->gather_summaries = (int (*)(bt_comm *, const uint32_t *, uint64_t, uint32_t *, uint64_t, uint64_t *))sym("bt_comm_gather_summaries");
    c->api->allgatherv = (int (*)(bt_comm *, const uint8_t *, uint64_t, uint8_t *, uint64_t, uint64_t *))sym("bt_comm_allgatherv");
    uint8_t id[BT_COMM_ID_BYTES];
    if (rank == 0) {
        check(c->api->unique_id(id), "bt_comm_unique_id");
        const std::string tmp = std::string(id_file) + ".tmp";
        {
            std::ofstream f(tmp, std::ios::binary);
            f.write((const char *)id, BT_COMM_ID_BYTES);
            f.write(nonce.data(), (std::streamsize)nonce.size());
            if (!f) throw std::runtime_error("cannot write " + tmp);
        }
        if (std::rename(tmp.c_str(), id_file) != 0) throw std::runtime_error(std::string("cannot write ") + id_file);   // (appears atomically)
    } else {
        bool got = false;
        for (int tries = 0; tries < 6000 && !got; ++tries) {   // up to ten minutes: rank 0 may still be reading its inputs
            std::ifstream f(id_file, std::ios::binary);
            if (f && f.read((char *)id, BT_COMM_ID_BYTES)) {
                const std::string rest((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
                got = rest == nonce;   // (another run's id: keep waiting for this run's)
            }
            if (!got) std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
        if (!got) throw std::runtime_error(std::string("rank ") + std::to_string(rank) + ": no communicator id in " + id_file);
    }
    check(c->api->init(ctx, id, rank, world, &c->comm), "bt_comm_init");
    return c;
}

Comm::Comm() {}

Comm::~Comm() {
    if (comm && api) api->destroy(comm);
    if (dl) dlclose(dl);
    // the run is over: rank 0 takes its rendezvous file away (a launcher of one's own does not have to); with RCCL every rank has joined the communicator by
    // now, with the files transport the directory protocol below waits for all ranks first
    if (rank_ == 0 && !id_file_.empty() && dir.empty()) std::remove(id_file_.c_str());
    if (!dir.empty()) {
        // A rank's file of the LAST exchange may still be unread by a slower rank, so nobody removes its own: every rank leaves a "done"
        // marker when it is through, and rank 0 clears the directory once all markers are there (or after a while, if a rank died).
        { std::ofstream f(dir + "/done." + std::to_string(rank_)); }
        if (rank_ == 0) {
            for (int tries = 0; tries < 3000; ++tries) {
                int done = 0;
                for (int r = 0; r < world_; r++) done += access((dir + "/done." + std::to_string(r)).c_str(), F_OK) == 0 ? 1 : 0;
                if (done == world_) break;
                std::this_thread::sleep_for(std::chrono::milliseconds(10));
            }
            for (int r = 0; r < world_; r++) {
                std::remove((dir + "/done." + std::to_string(r)).c_str());
                for (uint64_t s = seq > 2 ? seq - 2 : 0; s <= seq; s++) std::remove((dir + "/" + std::to_string(s) + "." + std::to_string(r)).c_str());
            }
            rmdir(dir.c_str());
            if (!id_file_.empty()) std::remove(id_file_.c_str());
        }
    }
}

// files transport: exchange number `seq` — every rank writes <dir>/<seq>.<rank> (appearing atomically), then reads all ranks' files.  Once
// a rank has read everybody's file of exchange n, everybody has finished exchange n - 1, so its own file of exchange n - 1 can go.
std::vector<std::vector<uint8_t>> Comm::exchangeFiles(const void *mine, size_t bytes) {
    const uint64_t n = seq++;
    const std::string base = dir + "/" + std::to_string(n) + ".";
    {
        const std::string tmp = base + std::to_string(rank_) + ".tmp";
        std::ofstream f(tmp, std::ios::binary);
        if (bytes) f.write((const char *)mine, (std::streamsize)bytes);
        f.close();
        if (!f || std::rename(tmp.c_str(), (base + std::to_string(rank_)).c_str()) != 0) throw std::runtime_error("files transport: cannot write " + tmp);
    }
    std::vector<std::vector<uint8_t>> all((size_t)world_);
    for (int r = 0; r < world_; r++) {
        const std::string name = base + std::to_string(r);
        bool got = false;
        for (int tries = 0; tries < 360000 && !got; ++tries) {
            std::ifstream f(name, std::ios::binary | std::ios::ate);
            if (f) {
                const std::streamsize len = f.tellg();
                all[r].resize((size_t)len);
                f.seekg(0);
                if (len == 0 || f.read((char *)all[r].data(), len)) got = true;
            }
            if (!got) {
                if (tries % 20 == 19 && access((dir.substr(0, dir.size() - 2) + ".failed").c_str(), F_OK) == 0) throw std::runtime_error("files transport: another rank of this run failed");
                std::this_thread::sleep_for(std::chrono::milliseconds(tries < 100 ? 1 : 10));
            }
        }
        if (!got) throw std::runtime_error("files transport: rank " + std::to_string(r) + " did not reach exchange " + std::to_string(n));
    }
    if (n > 0) std::remove((dir + "/" + std::to_string(n - 1) + "." + std::to_string(rank_)).c_str());
    return all;
}

void Comm::allreduceHist(uint64_t *hist, size_t n) {
    if (!dir.empty()) {
        const std::vector<std::vector<uint8_t>> all = exchangeFiles(hist, n * 8);
        std::vector<uint64_t> sum(n, 0);
        for (auto &part : all) {
            if (part.size() != n * 8) throw std::runtime_error("files transport: all-reduce of different lengths");
            for (size_t i = 0; i < n; i++) {
                uint64_t v;
                std::memcpy(&v, part.data() + i * 8, 8);
                sum[i] += v;
            }
        }
        std::memcpy(hist, sum.data(), n * 8);
        return;
    }
    DeviceBuffer d(ctx, n * 8);
    check(bt_memcpy_h2d(ctx, d.p, hist, n * 8), "bt_memcpy_h2d");
    check(api->allreduce_hist(comm, (uint64_t *)d.p, n), "bt_comm_allreduce_hist");
    check(bt_sync(ctx), "bt_sync");
    check(bt_memcpy_d2h(ctx, hist, d.p, n * 8), "bt_memcpy_d2h");
}

void Comm::allreduceDeviceAsync(uint64_t *d_hist, size_t n) {
    if (!dir.empty()) throw std::runtime_error("Comm: the files transport has no device-side reduction");
    check(api->allreduce_hist(comm, d_hist, n), "bt_comm_allreduce_hist");
}

void Comm::barrier() {
    uint64_t one = 1;
    allreduceHist(&one, 1);
}

std::vector<uint64_t> Comm::allgatherDevice(const uint8_t *d_local, uint64_t local_bytes, uint8_t *d_out, uint64_t capacity) {
    std::vector<uint64_t> off((size_t)world_ + 1);
    if (!dir.empty()) {
        std::vector<uint8_t> mine(local_bytes);
        check(bt_sync(ctx), "bt_sync");
        if (local_bytes) check(bt_memcpy_d2h(ctx, mine.data(), d_local, local_bytes), "bt_memcpy_d2h");
        const std::vector<std::vector<uint8_t>> all = exchangeFiles(mine.data(), mine.size());
        off[0] = 0;
        for (int r = 0; r < world_; r++) off[r + 1] = off[r] + all[r].size();
        if (off[world_] > capacity) throw std::runtime_error("allgatherDevice: output buffer too small");
        for (int r = 0; r < world_; r++)
            if (!all[r].empty()) check(bt_memcpy_h2d(ctx, d_out + off[r], all[r].data(), all[r].size()), "bt_memcpy_h2d");
        return off;
    }
    check(api->allgatherv(comm, d_local, local_bytes, d_out, capacity, off.data()), "bt_comm_allgatherv");
    check(bt_sync(ctx), "bt_sync");
    return off;
}

std::vector<uint8_t> Comm::allgatherBytes(const std::vector<uint8_t> &mine, std::vector<uint64_t> *offsets) {
    if (!dir.empty()) {
        const std::vector<std::vector<uint8_t>> parts = exchangeFiles(mine.data(), mine.size());
        std::vector<uint8_t> all;
        std::vector<uint64_t> off(1, 0);
        for (auto &p : parts) {
            all.insert(all.end(), p.begin(), p.end());
            off.push_back(all.size());
        }
        if (offsets) *offsets = off;
        return all;
    }
    // sizes first (one all-reduce of a world-long vector), so that the receive buffer can be sized
    std::vector<uint64_t> sizes((size_t)world_, 0);
    sizes[rank_] = mine.size();
    allreduceHist(sizes.data(), sizes.size());
    const uint64_t total = std::accumulate(sizes.begin(), sizes.end(), (uint64_t)0);
    DeviceBuffer d_in(ctx, mine.size()), d_out(ctx, total);
    if (!mine.empty()) check(bt_memcpy_h2d(ctx, d_in.p, mine.data(), mine.size()), "bt_memcpy_h2d");
    const std::vector<uint64_t> off = allgatherDevice((const uint8_t *)d_in.p, mine.size(), (uint8_t *)d_out.p, std::max<uint64_t>(total, 16));
    std::vector<uint8_t> all(total);
    if (total) check(bt_memcpy_d2h(ctx, all.data(), d_out.p, total), "bt_memcpy_d2h");
    if (offsets) *offsets = off;
    return all;
}

std::vector<uint32_t> Comm::gatherWords(const std::vector<uint32_t> &mine, std::vector<uint64_t> *offsets) {
    if (!dir.empty()) {
        const std::vector<std::vector<uint8_t>> parts = exchangeFiles(mine.data(), mine.size() * 4);
        std::vector<uint32_t> all;
        std::vector<uint64_t> off(1, 0);
        for (auto &p : parts) {
            if (rank_ == 0) {
                const size_t at = all.size();
                all.resize(at + p.size() / 4);
                if (!p.empty()) std::memcpy(all.data() + at, p.data(), p.size());
            }
            off.push_back(off.back() + p.size() / 4);
        }
        if (offsets) *offsets = off;
        return all;
    }
    std::vector<uint64_t> sizes((size_t)world_, 0);
    sizes[rank_] = mine.size();
    allreduceHist(sizes.data(), sizes.size());
    const uint64_t total = std::accumulate(sizes.begin(), sizes.end(), (uint64_t)0);
    DeviceBuffer d_in(ctx, mine.size() * 4), d_out(ctx, rank_ == 0 ? total * 4 : 16);
    if (!mine.empty()) check(bt_memcpy_h2d(ctx, d_in.p, mine.data(), mine.size() * 4), "bt_memcpy_h2d");
    std::vector<uint64_t> off((size_t)world_ + 1);
    check(api->gather_summaries(comm, (const uint32_t *)d_in.p, mine.size(), (uint32_t *)d_out.p, rank_ == 0 ? std::max<uint64_t>(total, 4) : 0, off.data()), "bt_comm_gather_summaries");
    check(bt_sync(ctx), "bt_sync");
    std::vector<uint32_t> all;
    if (rank_ == 0) {
        all.resize(total);
        if (total) check(bt_memcpy_d2h(ctx, all.data(), d_out.p, total * 4), "bt_memcpy_d2h");
    }
    if (offsets) *offsets = off;
    return all;
}

std::vector<uint32_t> Comm::gatherWordsDevice(const uint32_t *d_mine, uint64_t num_words, std::vector<uint64_t> *offsets) {
    if (!dir.empty()) {   // (files transport: the string leaves the device here, as a rank's part of the exchange)
        std::vector<uint32_t> mine(num_words);
        if (num_words) check(bt_memcpy_d2h(ctx, mine.data(), d_mine, num_words * 4), "bt_memcpy_d2h");
        return gatherWords(mine, offsets);
    }
    std::vector<uint64_t> sizes((size_t)world_, 0);
    sizes[rank_] = num_words;
    allreduceHist(sizes.data(), sizes.size());
    const uint64_t total = std::accumulate(sizes.begin(), sizes.end(), (uint64_t)0);
    DeviceBuffer d_none(ctx, 16), d_out(ctx, rank_ == 0 ? total * 4 : 16);
    std::vector<uint64_t> off((size_t)world_ + 1);
    check(api->gather_summaries(comm, num_words ? d_mine : (const uint32_t *)d_none.p, num_words, (uint32_t *)d_out.p, rank_ == 0 ? std::max<uint64_t>(total, 4) : 0, off.data()),
          "bt_comm_gather_summaries");
    check(bt_sync(ctx), "bt_sync");
    std::vector<uint32_t> all;
    if (rank_ == 0) {
        all.resize(total);
        if (total) check(bt_memcpy_d2h(ctx, all.data(), d_out.p, total * 4), "bt_memcpy_d2h");
    }
    if (offsets) *offsets = off;
    return all;
}

DeviceWords::~DeviceWords() {
    if (p) bt_free(ctx, p);
}
void DeviceWords::append(const uint32_t *d_words, uint64_t num_words) {
    if (n + num_words > cap) {
        const uint64_t want = std::max<uint64_t>(n + num_words, cap + cap / 2);
        void *q = nullptr;
        check(bt_malloc(ctx, std::max<uint64_t>(want, 4) * 4, &q), "bt_malloc");
        if (n) check(bt_memcpy_d2d(ctx, q, p, n * 4), "bt_memcpy_d2d");
        if (p) bt_free(ctx, p);
        p = q;
        cap = want;
    }
    if (num_words) check(bt_memcpy_d2d(ctx, (uint32_t *)p + n, d_words, num_words * 4), "bt_memcpy_d2d");
    n += num_words;
    k += 1;
}

std::vector<std::vector<uint32_t>> assignGroups(const GibbsBatchData &unit, int world) {
    const uint32_t G = unit.numGroups();
    std::vector<double> cost(G, 0.0);
    for (uint32_t g = 0; g < G; g++)
        for (uint32_t c = unit.group_cluster_off[g]; c < unit.group_cluster_off[g + 1]; c++) {
            const double H = unit.num_haplotypes[c], K = unit.kmer_off[c + 1] - unit.kmer_off[c];
            cost[g] += H * (H + 1) / 2 + 8.0 + K / 16.0;
        }
    std::vector<uint32_t> order(G);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return cost[a] > cost[b]; });
    std::vector<std::vector<uint32_t>> ids((size_t)world);
    for (uint32_t pos = 0; pos < G; pos++) {
        const uint32_t rnd = pos / world, col = pos % world;
        ids[rnd % 2 == 0 ? col : world - 1 - col].push_back(order[pos]);
    }
    for (auto &v : ids) std::sort(v.begin(), v.end());
    return ids;
}

// wire format of a launch's results (32-bit words; bt_gibbs_result_words builds the same string on the device): C, nd, nc, S, then per cluster
// (entries, cells), then h1 | h2 << 16 per entry, the frequencies (nd * S), a pad word if the count so far is odd, the statistics (nc * 12
// doubles as word pairs).  A rank's string is its launches' strings one after the other.
namespace {
BatchResults rebuildOnRankZero(const std::vector<uint32_t> &all, const std::vector<uint64_t> &off, int world, const GibbsBatchData &unit, const std::vector<std::vector<uint32_t>> &ids,
                               uint32_t S) {
    // where every cluster of the unit sits: (rank, cluster index within the rank's string)
    const uint32_t C = unit.numClusters();
    std::vector<uint32_t> c_rank(C, 0), c_local(C, 0);
    for (int r = 0; r < world; r++) {
        uint32_t local = 0;
        for (uint32_t g : ids[r])
            for (uint32_t c = unit.group_cluster_off[g]; c < unit.group_cluster_off[g + 1]; c++) {
                c_rank[c] = (uint32_t)r;
                c_local[c] = local++;
            }
    }
    struct Part {   // one launch of one rank
        const uint32_t *sizes, *keys, *freq;
        const uint8_t *stats;
        uint32_t first = 0, count = 0;   // the rank-local cluster indices it holds
        std::vector<uint64_t> dip_off, cell_off;
    };
    std::vector<std::vector<Part>> parts((size_t)world);
    std::vector<std::vector<uint32_t>> part_of((size_t)world);   // rank-local cluster index -> launch
    for (int r = 0; r < world; r++) {
        uint64_t at = off[r];
        uint32_t first = 0;
        while (at < off[r + 1]) {
            if (off[r + 1] - at < 4) throw std::runtime_error("gatherResults: truncated result string from rank " + std::to_string(r));
            const uint32_t *p = all.data() + at;
            const uint32_t Cr = p[0], ndr = p[1], ncr = p[2];
            if (p[3] != S) throw std::runtime_error("gatherResults: rank " + std::to_string(r) + " sampled another number of samples");
            const uint64_t at_stats = (4 + 2 * (uint64_t)Cr + (uint64_t)ndr * (1 + S) + 1) & ~1ull, words = at_stats + (uint64_t)ncr * 24;
            if (off[r + 1] - at < words) throw std::runtime_error("gatherResults: truncated result string from rank " + std::to_string(r));
            Part P;
            P.sizes = p + 4;
            P.keys = P.sizes + 2 * (size_t)Cr;
            P.freq = P.keys + ndr;
            P.stats = (const uint8_t *)(p + at_stats);
            P.first = first;
            P.count = Cr;
            P.dip_off.assign((size_t)Cr + 1, 0);
            P.cell_off.assign((size_t)Cr + 1, 0);
            for (uint32_t c = 0; c < Cr; c++) {
                P.dip_off[c + 1] = P.dip_off[c] + P.sizes[2 * c];
                P.cell_off[c + 1] = P.cell_off[c] + P.sizes[2 * c + 1];
            }
            if (P.dip_off[Cr] != ndr || P.cell_off[Cr] != ncr) throw std::runtime_error("gatherResults: inconsistent result string from rank " + std::to_string(r));
            part_of[r].insert(part_of[r].end(), Cr, (uint32_t)parts[r].size());
            parts[r].push_back(std::move(P));
            first += Cr;
            at += words;
        }
    }
    BatchResults full;
    full.dip_off.assign((size_t)C + 1, 0);
    full.cell_off.assign((size_t)C + 1, 0);
    uint64_t nd = 0, nc = 0;
    for (uint32_t c = 0; c < C; c++) {
        if (c_local[c] >= part_of[c_rank[c]].size()) throw std::runtime_error("gatherResults: rank " + std::to_string(c_rank[c]) + " did not send all its clusters");
        const Part &P = parts[c_rank[c]][part_of[c_rank[c]][c_local[c]]];
        const uint32_t l = c_local[c] - P.first;
        nd += P.dip_off[l + 1] - P.dip_off[l];
        nc += P.cell_off[l + 1] - P.cell_off[l];
        full.dip_off[c + 1] = nd;
        full.cell_off[c + 1] = nc;
    }
    full.h1.resize(std::max<uint64_t>(nd, 1));
    full.h2.resize(std::max<uint64_t>(nd, 1));
    full.freq.resize(std::max<uint64_t>(nd * S, 1));
    full.stats.resize(std::max<uint64_t>(nc * 12, 1));
    // (every cluster's rows go to their own place: the -p host threads — BT_HOST_THREADS, set by the executable — share the clusters)
    unsigned threads = 1;
    if (const char *e = getenv("BT_HOST_THREADS")) threads = (unsigned)std::max(1, atoi(e));
    parallelFor(C, threads, [&](size_t c_begin, size_t c_end, unsigned) {
        for (size_t c = c_begin; c < c_end; c++) {
            const Part &P = parts[c_rank[c]][part_of[c_rank[c]][c_local[c]]];
            const uint32_t l = c_local[c] - P.first;
            const uint64_t e0 = P.dip_off[l], e1 = P.dip_off[l + 1], k0 = P.cell_off[l], k1 = P.cell_off[l + 1];
            uint64_t to = full.dip_off[c];
            for (uint64_t e = e0; e < e1; e++, to++) {
                full.h1[to] = (uint16_t)(P.keys[e] & 0xFFFFu);
                full.h2[to] = (uint16_t)(P.keys[e] >> 16);
            }
            if (e1 > e0) std::memcpy(full.freq.data() + full.dip_off[c] * S, P.freq + e0 * S, (e1 - e0) * S * 4);
            if (k1 > k0) std::memcpy(full.stats.data() + full.cell_off[c] * 12, P.stats + k0 * 96, (k1 - k0) * 96);
        }
    });
    return full;
}
}  // namespace

BatchResults gatherResults(Comm &comm, const GibbsBatchData &unit, const std::vector<std::vector<uint32_t>> &ids, const BatchResults &mine, uint32_t S) {
    const uint32_t Cm = mine.dip_off.empty() ? 0u : (uint32_t)mine.dip_off.size() - 1;
    const uint64_t nd = Cm ? mine.dip_off[Cm] : 0, nc = Cm ? mine.cell_off[Cm] : 0;
    if (nd >> 32 || nc >> 32) throw std::runtime_error("gatherResults: more than 2^32 entries on one rank");
    std::vector<uint32_t> w;
    if (Cm) {   // (a rank without clusters sends nothing)
        w.reserve(5 + 2 * (size_t)Cm + nd * (1 + S) + nc * 24);
        w.push_back(Cm);
        w.push_back((uint32_t)nd);
        w.push_back((uint32_t)nc);
        w.push_back(S);
        for (uint32_t c = 0; c < Cm; c++) {
            w.push_back((uint32_t)(mine.dip_off[c + 1] - mine.dip_off[c]));
            w.push_back((uint32_t)(mine.cell_off[c + 1] - mine.cell_off[c]));
        }
        for (uint64_t e = 0; e < nd; e++) w.push_back((uint32_t)mine.h1[e] | ((uint32_t)mine.h2[e] << 16));
        for (uint64_t i = 0; i < nd * S; i++) w.push_back(mine.freq[i]);
        if (w.size() & 1) w.push_back(0);
        const size_t at = w.size();
        w.resize(at + nc * 24);
        if (nc) std::memcpy(w.data() + at, mine.stats.data(), nc * 96);
    }
    std::vector<uint64_t> off;
    const std::vector<uint32_t> all = comm.gatherWords(w, &off);
    if (comm.rank() != 0) return BatchResults();
    return rebuildOnRankZero(all, off, comm.world(), unit, ids, S);
}

BatchResults gatherResults(Comm &comm, const GibbsBatchData &unit, const std::vector<std::vector<uint32_t>> &ids, const DeviceWords &mine, uint32_t S) {
    std::vector<uint64_t> off;
    const std::vector<uint32_t> all = comm.gatherWordsDevice(mine.data(), mine.size(), &off);
    if (comm.rank() != 0) return BatchResults();
    return rebuildOnRankZero(all, off, comm.world(), unit, ids, S);
}

}  // namespace bthost
