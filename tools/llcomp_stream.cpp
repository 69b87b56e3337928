// llcomp_stream [frames] [width] [height] [tile_w] [tile_h] [depth] [encodes_in_flight] [frames_per_job] [pipelines] [--devices a,b,...]
//
// --devices: every pipeline object is a dealer over these GPUs (llcomp_mi_stream_create_multi: one pipeline of `depth` slots per
// device, jobs dealt round-robin, results in submission order) -- BASELINE config 5's "round-robin over the GPUs" from one process.
//
// BASELINE config 5 driven from C++ through the C ABI alone (include/llcomp_mi.h, llcomp_mi_stream_*): `frames` distinct
// RGB8 noise frames stream host -> GPU -> host (SLICED container) -> GPU -> host with `depth` pipeline slots; every decoded
// frame is compared with its source (memcmp on worker threads); prints one JSON line with the steady-state rate (first 4
// frames of every pipeline excluded) and the compression ratio.  The reference's counterpart is a loop of llcompc / llcompd
// runs, one image per process (llcompc.cpp:25-41, llcompd.cpp:17-31).  This is also the example INTEGRATION.md points at for
// the pipeline: pinned source buffers, back-pressure (LLCOMP_MI_BUSY), an encode result handed to submit_decode as it is,
// and `pipelines` stream objects driven by a thread each (one pipeline returns its results in submission order and leaves
// a DMA direction idle now and then; two keep both busy).
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "../include/llcomp_mi.h"

namespace {

double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

void fill_noise(uint8_t* p, size_t n, uint64_t seed) {  // xorshift64*: incompressible bytes, different per frame
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) {
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const uint64_t v = x * 0x2545F4914F6CDD1Dull;
        std::memcpy(p + i, &v, 8);
    }
    for (; i < n; ++i) p[i] = uint8_t(x >> (8 * (i & 7)));
}

struct Check {  // a decoded frame waiting for its comparison; the slot is released by the driving thread afterwards
    uint32_t slot;
    uint64_t tag;
    const uint8_t* got;
};

struct Config {
    uint32_t w, h, c, tw, th, depth, max_enc, fpj;
    std::vector<int32_t> devices;  // empty: the current device
};

struct Outcome {
    int fail = 0;
    bool mismatch = false;
    uint64_t container_bytes = 0;
    uint32_t busy = 0;
    std::vector<double> done_at;  // completion time of every job, seconds since `origin`
};

// one pipeline: `n` jobs of cfg.fpj frames each, sources at src + raw * job
void drive(const Config& cfg, const uint8_t* src, uint32_t n, size_t raw, double origin, Outcome& out) {
    llcomp_mi_stream* st = nullptr;
    const int rc_create = cfg.devices.empty()
                              ? llcomp_mi_stream_create_ex(&st, -1, cfg.w, cfg.h, cfg.c, cfg.tw, cfg.th, 1, cfg.depth, cfg.fpj)
                              : llcomp_mi_stream_create_multi(&st, cfg.devices.data(), uint32_t(cfg.devices.size()), cfg.w, cfg.h, cfg.c, cfg.tw,
                                                              cfg.th, 1, cfg.depth, cfg.fpj);
    if (int rc = rc_create) {
        std::fprintf(stderr, "llcomp_mi_stream_create: %s\n", llcomp_mi_strerror(rc));
        out.fail = rc;
        return;
    }
    // comparison workers
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Check> todo, done;
    std::atomic<bool> quit{false}, mismatch{false};
    std::vector<std::thread> workers;
    for (int t = 0; t < 3; ++t)
        workers.emplace_back([&] {
            for (;;) {
                Check k;
                {
                    std::unique_lock<std::mutex> lock(mu);
                    cv.wait(lock, [&] { return quit || !todo.empty(); });
                    if (todo.empty()) return;
                    k = todo.front();
                    todo.pop_front();
                }
                if (std::memcmp(k.got, src + raw * k.tag, raw) != 0) mismatch = true;
                std::lock_guard<std::mutex> lock(mu);
                done.push_back(k);
            }
        });

    out.done_at.assign(n, 0.0);
    std::vector<llcomp_mi_stream_result> enc_held(n);  // encode results whose containers a decode job still reads
    std::deque<llcomp_mi_stream_result> to_decode;
    uint32_t next = 0, finished = 0, enc_in_flight = 0;
    int fail = 0;
    const uint32_t fpj = cfg.fpj;
    while (finished < n && !fail) {
        bool progressed = false;
        {  // frames whose comparison is done: give their slots back
            std::lock_guard<std::mutex> lock(mu);
            while (!done.empty()) {
                llcomp_mi_stream_release(st, done.front().slot);
                done.pop_front();
                ++finished;
                progressed = true;
            }
        }
        while (!to_decode.empty()) {  // containers first: their decode frees two slots
            const llcomp_mi_stream_result& r = to_decode.front();
            std::vector<const uint8_t*> ptrs(fpj);
            std::vector<size_t> lens(fpj);
            for (uint32_t f = 0; f < fpj; ++f) {  // the job's containers, straight out of the encode result's pinned buffer
                uint64_t len = 0;
                if (llcomp_mi_stream_result_part(st, r.slot, f, &ptrs[f], &len)) { fail = LLCOMP_MI_BAD_ARGS; break; }
                lens[f] = size_t(len);
            }
            if (fail) break;
            const int rc = llcomp_mi_stream_submit_decode_batch(st, ptrs.data(), lens.data(), r.tag);
            if (rc == LLCOMP_MI_BUSY) { ++out.busy; break; }
            if (rc) { fail = rc; break; }
            enc_held[r.tag] = r;
            to_decode.pop_front();
            progressed = true;
        }
        while (!fail && next < n && enc_in_flight < cfg.max_enc && to_decode.empty()) {
            const int rc = llcomp_mi_stream_submit_encode(st, src + raw * next, next);
            if (rc == LLCOMP_MI_BUSY) { ++out.busy; break; }
            if (rc) { fail = rc; break; }
            ++next;
            ++enc_in_flight;
            progressed = true;
        }
        if (fail) break;
        if (llcomp_mi_stream_pending(st) > 0 && (!progressed || llcomp_mi_stream_poll(st) == LLCOMP_MI_OK)) {
            llcomp_mi_stream_result r;
            if (int rc = llcomp_mi_stream_wait(st, &r)) { fail = rc; break; }
            if (r.status) { fail = r.status; break; }
            if (r.kind == LLCOMP_MI_JOB_ENCODE) {
                --enc_in_flight;
                out.container_bytes += r.len;
                to_decode.push_back(r);
            } else {
                out.done_at[r.tag] = now() - origin;
                llcomp_mi_stream_release(st, enc_held[r.tag].slot);
                std::lock_guard<std::mutex> lock(mu);
                todo.push_back({r.slot, r.tag, r.data});
                cv.notify_one();
            }
        } else if (!progressed) {
            std::this_thread::sleep_for(std::chrono::microseconds(50));  // every slot is held by a frame being compared
        }
    }
    quit = true;
    cv.notify_all();
    for (auto& t : workers) t.join();
    llcomp_mi_stream_destroy(st);
    out.fail = fail;
    out.mismatch = mismatch;
}

}  // namespace

int main(int argc, char** argv) {
    Config cfg;
    for (int i = 1; i + 1 < argc; ++i)  // the one flag: taken out of argv, the rest stays positional
        if (std::strcmp(argv[i], "--devices") == 0) {
            for (const char* p = argv[i + 1]; *p;) {
                char* e = nullptr;
                const long v = std::strtol(p, &e, 10);
                if (e == p || v < 0) return 1;
                cfg.devices.push_back(int32_t(v));
                p = *e == ',' ? e + 1 : e;
                if (*e && *e != ',') return 1;
            }
            for (int j = i; j + 2 < argc; ++j) argv[j] = argv[j + 2];
            argc -= 2;
            break;
        }
    const uint32_t frames_total = argc > 1 ? uint32_t(std::atoi(argv[1])) : 64;
    cfg.w = argc > 2 ? uint32_t(std::atoi(argv[2])) : 3840;
    cfg.h = argc > 3 ? uint32_t(std::atoi(argv[3])) : 2160;
    cfg.c = 3;
    cfg.tw = argc > 4 ? uint32_t(std::atoi(argv[4])) : 480;
    cfg.th = argc > 5 ? uint32_t(std::atoi(argv[5])) : 1;
    cfg.depth = argc > 6 ? uint32_t(std::atoi(argv[6])) : 8;
    cfg.max_enc = argc > 7 ? uint32_t(std::atoi(argv[7])) : 3;
    cfg.fpj = argc > 8 ? uint32_t(std::atoi(argv[8])) : 1;
    const uint32_t pipelines = argc > 9 ? uint32_t(std::atoi(argv[9])) : 1;
    const size_t raw1 = size_t(cfg.w) * cfg.h * cfg.c;
    if (!frames_total || !raw1 || !cfg.fpj || !pipelines || frames_total % (cfg.fpj * pipelines)) return 1;
    const uint32_t jobs = frames_total / cfg.fpj, per = jobs / pipelines;  // jobs in all, jobs per pipeline
    const size_t raw = raw1 * cfg.fpj;                                     // bytes of one job's frames

    uint8_t* src = static_cast<uint8_t*>(llcomp_mi_host_alloc(raw * jobs));  // pinned: the H2D copies are plain DMA
    if (!src) {
        std::fprintf(stderr, "llcomp_mi_host_alloc failed\n");
        return 1;
    }
    for (uint32_t i = 0; i < jobs; ++i) fill_noise(src + raw * i, raw, 1234 + i);

    std::vector<Outcome> outs(pipelines);
    std::vector<std::thread> drivers;
    const double origin = now();
    for (uint32_t p = 0; p < pipelines; ++p)
        drivers.emplace_back([&, p] { drive(cfg, src + raw * per * p, per, raw, origin, outs[p]); });
    for (auto& t : drivers) t.join();
    llcomp_mi_host_free(src);

    uint64_t container_bytes = 0;
    uint32_t busy = 0;
    for (auto& o : outs) {
        if (o.fail || o.mismatch) {
            std::fprintf(stderr, "llcomp_stream: %s\n", o.fail ? llcomp_mi_strerror(o.fail) : "a decoded frame differs from its source");
            return 1;
        }
        container_bytes += o.container_bytes;
        busy += o.busy;
    }
    // steady state: from the moment the last pipeline has the jobs that hold its first 4 frames back, to the end
    const uint32_t skip = per * cfg.fpj > 8 ? (4 + cfg.fpj - 1) / cfg.fpj : 0;
    double begin = 0.0, end = 0.0;
    for (auto& o : outs) {
        if (skip) begin = std::max(begin, o.done_at[skip - 1]);
        end = std::max(end, o.done_at[per - 1]);
    }
    uint32_t counted = 0;
    for (auto& o : outs)
        for (double t : o.done_at) counted += t > begin;
    const double steady = double(counted) * cfg.fpj * cfg.w * cfg.h / 1e6 / (end - begin);
    std::printf("{\"frames\": %u, \"frames_per_job\": %u, \"pipelines\": %u, \"width\": %u, \"height\": %u, \"tile\": \"%ux%u\", \"depth\": %u, "
                "\"devices\": %u, \"steady_mpix_s\": %.1f, \"compression_ratio\": %.4f, \"backpressure_hits\": %u, \"verified\": true}\n",
                frames_total, cfg.fpj, pipelines, cfg.w, cfg.h, cfg.tw, cfg.th, cfg.depth, uint32_t(cfg.devices.empty() ? 1 : cfg.devices.size()), steady, double(raw) * jobs / double(container_bytes), busy);
    return 0;
}
