/*
 * rt_gpu.hip -- kernels and C ABI of librt_gpu.so (gfx950 only).
 *
 * Entry points are declared, with the reference function each one replaces,
 * in include/rt_gpu.h.  Device code lives in rt_trace.hpp; the host scene
 * graph is flattened by rt_flatten.cpp.
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <mutex>
#include <string>
#include <vector>

#include "rt_gpu.h"
#include "rt_flatten.h"
#include "rt_trace.hpp"

#define RT_BLOCK 256                 /* 4 waves per workgroup */
#define RT_STAT_WORDS 20
#define RT_TOTAL_WORDS 3              /* after the per-kind counters: rounds that had work, paths lost to a full list, passes that started paths */
#ifndef RT_SHADE_WEIGHT
#define RT_SHADE_WEIGHT 16            /* in-wave scheduler (job replay, finish kernel): weight of a lane that wants a
                                       * shade step, in 1/16ths of a lane that wants a tree step */
#endif

/* ------------------------------------------------------------------------ */
/* kernels                                                                   */
/* ------------------------------------------------------------------------ */

/* which 8x8 pixel blocks ("units") of a region one launch covers */
struct RenderParams {
    uint32_t blocks_x, block_rows;
    uint32_t band_blocks, band_rank, band_count;   /* 0 band_blocks = no split */
};

__device__ __forceinline__ uint32_t wave_sum(uint32_t v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool STATS>
__device__ __forceinline__ void flush_counters(const rt::Counters &cn, uint64_t *stats)
{
    if (!stats) return;
    const uint32_t *w = reinterpret_cast<const uint32_t *>(&cn);
    const int lane = threadIdx.x & 63;
    if (STATS) {
#pragma unroll
        for (int k = 0; k < RT_STAT_WORDS; k++) {
            uint32_t s = wave_sum(w[k]);
            if (lane == 0 && s) atomicAdd(reinterpret_cast<unsigned long long *>(stats + k), (unsigned long long)s);
        }
    } else {
        uint32_t s = wave_sum(cn.rays);
        if (lane == 0 && s) atomicAdd(reinterpret_cast<unsigned long long *>(stats + 1), (unsigned long long)s);
    }
}

/* ------------------------------------------------------------------------
 * The frame is rendered by two kernels that hand paths to each other through
 * HBM:
 *
 *   rt_shade_kernel   everything of the path loop except walking mesh trees.
 *       A lane carries a path through events (draw -> ray -> lights, spheres,
 *       planes, instance root boxes -> bookkeeping) until the path ends (its
 *       value goes to the sample scratch) or its ray enters an instance's root
 *       box; then the lane writes the path as a 160-byte record to the parked
 *       list and takes the next path: it never waits for a tree walk.
 *   rt_tree_kernel    walks the trees for the parked rays only (6 % of the
 *       rays of the benchmark frame, ~40 node/triangle steps each): small
 *       register footprint, many waves per SIMD to hide the dependent fetches,
 *       lanes take the next parked ray as soon as theirs is done.
 *
 * Round 0 shades fresh (pixel, sample) items; round r > 0 resumes the paths
 * parked in round r-1 after the tree kernel has answered their queries.  The
 * loop ends when a shade launch parks nothing, or with rt_finish_kernel once
 * few paths are left.  A path executes exactly the
 * operations of the host's nested loops in the same order whatever lane or
 * launch runs them, so results do not change.
 * ------------------------------------------------------------------------ */
#ifndef RT_TREE_REFILL
#define RT_TREE_REFILL 16             /* tree kernel: free lanes needed before the wave fetches new rays */
#endif
#ifndef RT_TREE_SCAN_MIN
#define RT_TREE_SCAN_MIN 12           /* tree kernel: lanes waiting to test their next instance's root box that make the test worth running */
#endif
#ifndef RT_TREE_BOTH
#define RT_TREE_BOTH 12               /* tree kernel: lanes that make the minority kind of step worth running too */
#endif
#ifndef RT_SMALL_CHUNK
#define RT_SMALL_CHUNK 64u              /* the same in rounds with few paths (less padding for the next launch to skip) */
#endif
#define RT_CHUNK 256u                /* parked-list slots a wave reserves per atomic */

/* One record per round of a batch, in device memory, all zeroed by one memset before the batch's first
 * launch.  Round r's shade launch fills out_reserved / parked / overflow of record r; the tree launch
 * and the resumed-path launch of round r+1 (or the finishing launch) read them there and advance their
 * own cursors in the same record, so no launch needs a value from the host: the host enqueues a fixed
 * sequence of launches and the ones with nothing to do return at once. */
struct QueueCounters {
    uint32_t unit_next;              /* new paths: next portion of fresh work (the batch's cursor lives in its first record) */
    uint32_t in_next;                /* cursor of whoever carries this round's records on (resumed-path or finishing launch) */
    uint32_t out_reserved;           /* slots reserved in this round's output list */
    uint32_t overflow;               /* paths that found the output list full (cannot happen: see ShadeParams.out_limit) */
    uint32_t parked;                 /* records actually parked (out_reserved counts whole chunks) */
    uint32_t tree_next;              /* cursor of the tree launch that answers this round's records */
    uint32_t pad[2];                 /* [0]: round 0 only: the pass found paths to start */
};

struct ShadeParams {
    rt_dev_scene S;
    float   *samples;                /* [sample of this launch][unit][pixel of the 8x8 block] RGB */
    uint64_t *stats;
    QueueCounters *q;                /* this round's record */
    QueueCounters *q_in;             /* resume: the previous round's record (its list is the input) */
    const uint32_t *in_list;         /* resume: records parked by the previous round (answered) */
    uint32_t *out_list;              /* records this launch parks */
    uint32_t out_capacity;           /* slots of either list */
    uint32_t *portion_next;          /* new paths: the batch's cursor over its portions of fresh work (shared by the passes of the batch) */
    uint32_t out_limit;              /* new paths, lists smaller than the batch: a wave takes no further portion once it has seen this many
                                      * slots reserved (a reservation returns the count of all waves; what the waves hold and take until
                                      * they reserve again still fits); the portions left wait for the next pass.  0: no limit */
    uint32_t finish_below;           /* resume: the round runs only if more records than this were parked (else the finishing launch takes them) */
    uint32_t *more_passes;           /* new paths, second and later passes of a batch: set to 1 (host memory) when the pass finds paths to start,
                                      * i.e. the lists were too small for one pass; NULL in a batch's first pass */
    uint64_t *pass_count;            /* new paths: += 1 per pass that finds paths to start */
    uint32_t chunk;                  /* new paths: output slots a wave reserves per atomic (<= RT_CHUNK, multiple of 64) */
    uint32_t tail_lanes;             /* a wave left with fewer paths than this and no input hands them to the next round (0: never) */
    uint32_t width, height;          /* image */
    uint32_t x0, y0, w, h;           /* region */
    uint32_t blocks_x, block_rows;   /* units = 8x8 pixel blocks of this launch */
    uint32_t band_blocks, band_rank, band_count;
    uint32_t first_sample, launch_spp, sample_offset, seed0, seed1;
    int      max_segments;
    float    half_w, half_h;
};

#ifndef RT_SLOTS
#define RT_SLOTS 56u                  /* LDS record slots of a resumed-path wave: with 56 (not 64) four workgroups fit the LDS of a CU */
#endif
#define RT_SLOT_MASK (RT_SLOTS >= 64u ? ~0ull : (1ull << (RT_SLOTS & 63u)) - 1ull)
#define RT_STAGE_WORDS (RT_SLOTS * 40u + 64u)   /* LDS words of a resumed-path wave: the record slots of 10 quads + 64 slot numbers */
#define RT_STAGE_LDS ((size_t)(RT_BLOCK / 64) * RT_STAGE_WORDS * 4u)
#ifndef RT_STAGE_REFILL
#define RT_STAGE_REFILL 16            /* empty record slots that make a resumed-path wave fetch more of the list */
#endif
#define RT_POOL 128u                  /* first hits of new paths a new-path wave can hold in LDS (3 quads each) */
#define RT_POOL_LDS ((size_t)(RT_BLOCK / 64) * RT_POOL * 48u)
#ifndef RT_IN_BATCH
#define RT_IN_BATCH 256u              /* list slots a resumed-path wave takes per atomic on the shared cursor */
#endif
#ifndef RT_TREE_IN_BATCH
#define RT_TREE_IN_BATCH 128u         /* the same for a tree-kernel wave */
#endif
#ifndef RT_PORTION_SPP
#define RT_PORTION_SPP 8u             /* samples per pixel in one portion of fresh work (x 64 pixels) */
#endif
#ifndef RT_SHADE_CLASSES
#define RT_SHADE_CLASSES 1            /* new paths: 1: per iteration only the lanes of the most wanted kind of step run;
                                       * 2: the two kinds that book a light sample count as one; 0: all lanes run */
#endif
#ifndef RT_RESUME_CLASSES
#define RT_RESUME_CLASSES RT_SHADE_CLASSES   /* the same for resumed paths */
#endif
#define RT_KIND_CLASS(fresh, k) (((fresh) ? RT_SHADE_CLASSES : RT_RESUME_CLASSES) == 0 ? 1 : \
                                 ((fresh) ? RT_SHADE_CLASSES : RT_RESUME_CLASSES) == 2 && (k) == 3 ? 2 : (k))
#ifndef RT_SHADE_WAVES
#define RT_SHADE_WAVES 4              /* waves per SIMD the shade kernel is compiled for (0: compiler's choice) */
#endif
#ifndef RT_RESUME_WAVES
#define RT_RESUME_WAVES RT_SHADE_WAVES
#endif
#if RT_SHADE_WAVES
#define RT_SHADE_ATTR __attribute__((amdgpu_waves_per_eu(FRESH ? RT_SHADE_WAVES : RT_RESUME_WAVES, FRESH ? RT_SHADE_WAVES : RT_RESUME_WAVES)))
#else
#define RT_SHADE_ATTR
#endif

template <bool FRESH, bool STATS, bool SPECULAR>
__global__ void __launch_bounds__(RT_BLOCK) RT_SHADE_ATTR
rt_shade_kernel(const ShadeParams P)
{
    /* Resumed paths: what there is to do stands in the previous round's record.  Few records: the
     * finishing launch takes them (this launch returns); otherwise as many workgroups as there are
     * 256-slot portions of the list work, the others return. */
    uint32_t in_slots = 0, chunk = P.chunk;
    if (!FRESH) {
        const uint32_t reserved = __builtin_amdgcn_readfirstlane(P.q_in->out_reserved);
        const uint32_t records = __builtin_amdgcn_readfirstlane(P.q_in->parked);
        if (records <= P.finish_below) return;
        in_slots = reserved < P.out_capacity ? reserved : P.out_capacity;
        if (blockIdx.x >= (in_slots + 255u) / 256u) return;
        /* few records per wave: reserve output slots in smaller chunks (less padding for the next launch to skip) */
        chunk = records >= (4u << 20) ? RT_CHUNK : RT_SMALL_CHUNK;
    } else if (__builtin_amdgcn_readfirstlane(*P.portion_next) >= P.blocks_x * P.block_rows * ((P.launch_spp + RT_PORTION_SPP - 1u) / RT_PORTION_SPP)) {
        return;                                          /* an earlier pass started every path of the batch */
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {    /* the launch's first workgroup sees the cursor before the launch has moved it far */
        P.q->pad[0] = 1u;                                /* this pass has work (the finishing launch counts its rounds) */
        if (P.pass_count) atomicAdd(reinterpret_cast<unsigned long long *>(P.pass_count), 1ull);
        if (P.more_passes) __hip_atomic_store(P.more_passes, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    rt::Stack stk; stk.base = nullptr; stk.stride = 0;                      /* this kernel walks no tree */
    rt::Counters cn; memset(&cn, 0, sizeof(cn));
    const uint32_t lane = threadIdx.x & 63;

    rt::PathMachine<rt::CounterRng, STATS, SPECULAR> pm(P.S, stk, cn);
    pm.half_w = P.half_w; pm.half_h = P.half_h; pm.max_segments = P.max_segments;
    pm.rng.seed0 = P.seed0; pm.rng.seed1 = P.seed1;
    pm.phase = rt::PH_DEAD;
    bool has = false;
    uint32_t sample_index = 0;

    /* wave-uniform cursors */
    /* fresh work comes in portions of one 8x8 pixel block x RT_PORTION_SPP samples: small enough to
     * keep thousands of waves evenly loaded on a fraction of a frame */
    const uint32_t units = P.blocks_x * P.block_rows;
    const uint32_t parts = (P.launch_spp + RT_PORTION_SPP - 1u) / RT_PORTION_SPP;
    uint32_t unit = 0, item_next = 0, unit_items = 0;   /* fresh: current block, next item and end of the portion */
    uint32_t bx = 0, by = 0;
    uint32_t in_cur = 0, in_end = 0;                    /* resume: slots taken from the shared cursor, not yet staged */
    bool cursor_done = false;
    bool input_done = false;
    extern __shared__ uint32_t lds[];
    /* Resumed paths: records come from the list through RT_SLOTS LDS slots per wave (slot j is filled by
     * lane j with global_load_lds: no registers, nothing to wait for at issue; quad q of slot j at
     * word (q * RT_SLOTS + j) * 4).  A record's kind of next step is known before a lane takes it, so a
     * free lane takes a record of the kind the wave is about to run: the lanes stay together. */
    uint32_t *stage = lds + (threadIdx.x >> 6) * RT_STAGE_WORDS;
    uint32_t *pick = stage + RT_SLOTS * 40u;                 /* slot numbers handed to the free lanes */
    unsigned long long ready = 0, flight = 0;           /* slots holding a record / being filled */
    int slot_kind = 0;                                  /* kind of the record in slot `lane` */
    auto stage_issue = [&]() {                           /* fill the empty slots from the list */
        const unsigned long long empty = ~(ready | flight) & RT_SLOT_MASK;
        if (in_cur >= in_end && !cursor_done) {
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&P.q_in->in_next, RT_IN_BATCH);
            in_cur = __builtin_amdgcn_readfirstlane(v);
            if (in_cur >= in_slots) { cursor_done = true; in_end = in_cur; }
            else in_end = in_cur + RT_IN_BATCH < in_slots ? in_cur + RT_IN_BATCH : in_slots;
        }
        const uint32_t room = (uint32_t)__popcll(empty);
        const uint32_t n = in_end - in_cur < room ? in_end - in_cur : room;
        const uint32_t rank = rt::bits_below_lane(empty);
        const bool mine = ((empty >> lane) & 1ull) && rank < n;
        if (mine) {
            const uint32_t *ray_words = rt::ray_part(P.in_list, in_cur + rank);
            const uint32_t *path_words = rt::path_part(P.in_list, P.out_capacity, in_cur + rank);
#pragma unroll
            for (uint32_t qd = 0; qd < 4u; qd++)
                __builtin_amdgcn_global_load_lds(ray_words + 4u * qd, stage + 4u * RT_SLOTS * qd, 16, 0, 0);
#pragma unroll
            for (uint32_t qd = 4; qd < 10u; qd++)
                __builtin_amdgcn_global_load_lds(path_words + 4u * (qd - 4u), stage + 4u * RT_SLOTS * qd, 16, 0, 0);
        }
        flight |= __ballot(mine);
        in_cur += n;
    };
    auto stage_land = [&]() {                            /* the slots being filled have their records: look at them */
        if (!flight) return;
        __builtin_amdgcn_s_waitcnt(0x0F70);              /* vmcnt(0) */
        bool ok = false;
        if ((flight >> lane) & 1ull) {
            const uint32_t f = stage[lane * 4u], w = stage[(3u * RT_SLOTS + lane) * 4u + 3u];
            const int ev = (int)((f >> 1) & 3u), ray = (int)((f >> 3) & 3u), li = (int)(w >> 16);
            slot_kind = ray == rt::RAY_SEGMENT ? 1 : (ev == rt::EV_NEE ? (li + 1 < P.S.light_count ? 2 : 3) : 0);
            slot_kind = RT_KIND_CLASS(false, slot_kind);
            ok = (f & 1u) != 0u;                         /* padding of the list: no record */
        }
        ready |= __ballot(ok);
        flight = 0;
    };
    if (!FRESH) { stage_issue(); stage_land(); stage_issue(); }
    uint32_t out_cur = 0, out_end = 0;                  /* reserved output range */
    uint32_t seen_reserved = 0;                         /* slots of the list reserved by all waves, as of this wave's latest reservation */
    uint32_t parked_here = 0;

    /* write the paths of the lanes that `want` it to the parked list (m: the machine holding them) */
    auto park = [&](bool want, auto &m, uint32_t sidx, bool ready_flag = false) {
        const unsigned long long parking = __ballot(want);
        if (!parking) return;
        const uint32_t count = (uint32_t)__popcll(parking);
        const uint32_t rank = rt::bits_below_lane(parking);
        const uint32_t room = out_end - out_cur;
        uint32_t next_base = 0;
        if (count > room) {                              /* reserve the next chunk; the old one is filled to its end */
            uint32_t v = 0;
            if (lane == 0) v = atomicAdd(&P.q->out_reserved, chunk);
            next_base = __builtin_amdgcn_readfirstlane(v);
            seen_reserved = next_base + chunk;
        }
        if (want) {
            uint32_t slot = rank < room ? out_cur + rank : next_base + (rank - room);
            if (slot < P.out_capacity) {
                m.store_record(rt::ray_part(P.out_list, slot), rt::path_part(P.out_list, P.out_capacity, slot), sidx, ready_flag);
            } else {                                      /* cannot happen: the host sizes the lists for every path of the launch */
                atomicAdd(&P.q->overflow, 1u);
            }
        }
        if (count > room) { out_cur = next_base + (count - room); out_end = next_base + chunk; }
        else out_cur += count;
        parked_here += count;
    };

    /* New paths (FRESH).  The first event of a path (camera ray, its test against the lights,
     * spheres and planes, the instance root boxes) is the same code for every path, so it runs for
     * 64 new paths at once, on all lanes, whatever the lanes' own paths are doing: their state
     * stays in its registers, the step works in temporaries.  A ray that enters a root box goes
     * to the parked list right away; the other paths wait in an LDS pool (pixel, direction, first
     * hit, generator key: 3 quads; quad q of entry i at pool[q * RT_POOL + i]) until a lane is free
     * to carry them on.  So the lanes' own steps only ever mix the three kinds of step of a path
     * under way, and those follow each other in a cycle that keeps the lanes together. */
    uint32_t pool_n = 0;
    uint4 *pool = reinterpret_cast<uint4 *>(lds) + (threadIdx.x >> 6) * (3u * RT_POOL);
    auto camera_step = [&]() {
        if (item_next >= unit_items) {                   /* next portion */
            if (P.out_limit && seen_reserved > P.out_limit) {
                input_done = true;                       /* the list is nearly full: the rest of the batch is another pass's */
                return;
            }
            uint32_t u = 0;
            if (lane == 0) u = atomicAdd(P.portion_next, 1u);
            u = __builtin_amdgcn_readfirstlane(u);
            if (u >= units * parts) { input_done = true; return; }
            unit = u / parts;
            const uint32_t first_spp = (u % parts) * RT_PORTION_SPP;
            const uint32_t end_spp = first_spp + RT_PORTION_SPP < P.launch_spp ? first_spp + RT_PORTION_SPP : P.launch_spp;
            item_next = 64u * first_spp; unit_items = 64u * end_spp;
            bx = unit % P.blocks_x; by = unit / P.blocks_x;
            if (P.band_blocks) {
                uint32_t k = by / P.band_blocks, in = by % P.band_blocks;
                by = (k * P.band_count + P.band_rank) * P.band_blocks + in;
            }
        }
        const uint32_t sample = item_next >> 6;          /* one sample of the block's 64 pixels per step */
        item_next += 64u;
        const uint32_t lx = bx * 8 + (lane & 7), ly = by * 8 + (lane >> 3);
        const uint32_t sidx = (sample * units + unit) * 64u + lane;
        rt::PathMachine<rt::CounterRng, STATS, SPECULAR> cam(P.S, stk, cn);
        bool to_park = false, to_pool = false;
        if (lx < P.w && ly < P.h) {
            cam.half_w = P.half_w; cam.half_h = P.half_h; cam.max_segments = P.max_segments;
            cam.rng.seed0 = P.seed0; cam.rng.seed1 = P.seed1;
            cam.x = P.x0 + lx; cam.y = P.y0 + ly;
            cam.start_path();
            cam.rng.begin(P.width * cam.y + cam.x, P.sample_offset + P.first_sample + sample);
            cam.n = cam.albedo = cam.direct = rt::mk(0, 0, 0);
            cam.reached_here = cam.shadow_len2 = cam.shadow_cos = 0;
            cam.li = 0;
            float tmax;
            (void)cam.event(tmax);                       /* the camera event never ends a path */
            cam.park_inst = -1;
            if (!rt::query_begin<STATS>(P.S, cam.o, cam.d, tmax, false, cam.h, cn))
                cam.park_inst = rt::query_probe<STATS>(P.S, cam.o, cam.d, cam.h, 0, cn);
            to_park = cam.park_inst >= 0;
            to_pool = !to_park;
        }
        park(to_park, cam, sidx);
        const unsigned long long pushing = __ballot(to_pool);
        if (to_pool) {
            const uint32_t e = pool_n + rt::bits_below_lane(pushing);
            pool[e] = make_uint4(cam.x | (cam.y << 16), sidx, __float_as_uint(cam.d.x), __float_as_uint(cam.d.y));
            pool[RT_POOL + e] = make_uint4(__float_as_uint(cam.d.z), __float_as_uint(cam.h.t), (uint32_t)cam.h.type, cam.h.id);
            pool[2u * RT_POOL + e] = make_uint4((uint32_t)cam.h.material, cam.rng.k0, cam.rng.k1, 0u);
        }
        pool_n += (uint32_t)__popcll(pushing);
    };

#if defined(RT_SHADE_CLOCKS)
    unsigned long long stick[16]; for (int i = 0; i < 16; i++) stick[i] = 0;
    stick[15] = __builtin_readcyclecounter();
    pm.ticks = stick;
    unsigned long long stick_steps = 0, stick_lanes = 0;
#endif
    for (;;) {
        RT_STICK(stick, 7)
        /* ---- 1. give free lanes a path ----------------------------------------- */
        if (FRESH) {
            const unsigned long long need = __ballot(!has);
            const uint32_t want = (uint32_t)__popcll(need);
            if (want) {
                while (pool_n < want && !input_done) camera_step();   /* always 64 new paths: the pool holds < 64 + 64 */
                const uint32_t take = want < pool_n ? want : pool_n;
                const uint32_t rank = rt::bits_below_lane(need);
                if (!has && rank < take) {
                    const uint32_t e = pool_n - 1u - rank;
                    const uint4 a = pool[e], b = pool[RT_POOL + e], c = pool[2u * RT_POOL + e];
                    pm.start_path();
                    pm.x = a.x & 0xFFFFu; pm.y = a.x >> 16;
                    sample_index = a.y;
                    pm.o = rt::ld3(P.S.cam.pos);
                    pm.d = rt::mk(__uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x));
                    pm.h.t = __uint_as_float(b.y); pm.h.type = (int)b.z; pm.h.id = b.w; pm.h.inst = 0;
                    pm.h.beta = 0.0f; pm.h.gamma = 0.0f; pm.h.material = (int)c.x; pm.h.entered = false;
                    pm.rng.k0 = c.y; pm.rng.k1 = c.z; pm.rng.c0 = 1u; pm.rng.c1 = 0u; pm.rng.fresh = false;
                    pm.ray = rt::RAY_SEGMENT;            /* the first hit is there to be booked */
                    has = true;
                }
                pool_n -= take;
            }
        }
        RT_STICK(stick, 0)
        /* ---- 2. one shade step ---------------------------------------------------- */
        /* Lanes about to do the same kind of step run together: a step's cost is dominated by the
         * code of its kind (book a surface hit / book a light sample, then a light-sample or bounce
         * event), and a wave pays for every kind that any active lane takes.  The kind that most
         * lanes want, or could want once free lanes have taken waiting records of that kind, runs;
         * the others keep their state and wait to become the majority (kinds follow each other in
         * a cycle, so waiting lanes are soon joined). */
        auto kind_of = [&]() {
            int kind = 0;                                 /* a path about to start (never in this kernel) */
            if (pm.ray == rt::RAY_SEGMENT) kind = 1;      /* book a surface hit, then usually the first light sample */
            else if (pm.ev == rt::EV_NEE) kind = (pm.li + 1 < P.S.light_count) ? 2 : 3;   /* next light sample / bounce */
            return RT_KIND_CLASS(FRESH, kind);
        };
        int best = 0;
        {
            const int kind = kind_of();
            int n[4], a[4] = {0, 0, 0, 0};
#pragma unroll
            for (int k = 0; k < 4; k++) n[k] = __popcll(__ballot(has && kind == k));
            if (!FRESH) {
                const int free_lanes = 64 - __popcll(__ballot(has));
                const bool slot_ready = (ready >> lane) & 1ull;
#pragma unroll
                for (int k = 1; k < 4; k++) {
                    a[k] = __popcll(__ballot(slot_ready && slot_kind == k));
                    if (a[k] > free_lanes) a[k] = free_lanes;
                }
            }
            int nb = n[0] + a[0];
#pragma unroll
            for (int k = 1; k < 4; k++) if (n[k] + a[k] > nb) { best = k; nb = n[k] + a[k]; }
            if (!FRESH) {
                if (nb == 0) {                            /* no path in a lane, no record waiting */
                    if (flight) { stage_land(); continue; }
                    if (cursor_done && in_cur >= in_end) break;
                    stage_issue();
                    continue;
                }
                if (a[best]) {                            /* free lanes take the waiting records of that kind */
                    const unsigned long long slots = __ballot(((ready >> lane) & 1ull) && slot_kind == best);
                    const unsigned long long need = __ballot(!has);
                    const uint32_t take = (uint32_t)a[best];
                    const uint32_t srank = rt::bits_below_lane(slots);
                    const bool given = ((slots >> lane) & 1ull) && srank < take;
                    if (given) pick[srank] = lane;
                    ready &= ~__ballot(given);
                    const uint32_t rank = rt::bits_below_lane(need);
                    if (!has && rank < take) {
                        const uint32_t j = pick[rank];
                        has = pm.load_record(reinterpret_cast<const uint4 *>(stage) + j, RT_SLOTS, sample_index);
                    }
                }
            }
        }
        if (FRESH && !__ballot(has)) break;              /* nothing in flight and no input left */
        const bool go = has && kind_of() == best;
        RT_STICK(stick, 1)
#if defined(RT_SHADE_CLOCKS)
        stick_steps++; stick_lanes += (unsigned long long)__popcll(__ballot(go));
#endif
        if (STATS) { if (lane == 0) cn.shade_steps++; if (go) cn.shade_lanes++; }
        if (go) {
            pm.shade_step_park([&]() {
                const rt::f3 c = pm.path_value();
                float *dst = P.samples + 3 * (size_t)sample_index;
                dst[0] = c.x; dst[1] = c.y; dst[2] = c.z;
                has = false;
                pm.phase = rt::PH_DEAD;
            });
        }

        RT_STICK(stick, 6)
        if (!FRESH) {                                    /* what was asked for during the last step has long arrived */
            stage_land();
            if (__popcll(~(ready | flight)) - (64 - (int)RT_SLOTS) >= RT_STAGE_REFILL && !(cursor_done && in_cur >= in_end)) stage_issue();
        }

        /* ---- 3. park the lanes whose ray has to walk a tree ------------------------- */
        const bool parks = has && pm.phase == rt::PH_PARK;
        park(parks, pm, sample_index);
        if (parks) { has = false; pm.phase = rt::PH_DEAD; }

        /* ---- 4. the end of a launch: a few long paths in a few lanes of a few waves.  They make a
         * better wave together with the next round's paths: a wave that has run out of input and
         * of most of its paths writes the rest to the list as they are (query complete, nothing
         * for the tree kernel) and ends.  The wave has just made a step, so every round advances. */
        if (P.tail_lanes) {
            const bool no_input = FRESH ? (input_done && pool_n == 0u)
                                        : (cursor_done && in_cur >= in_end && !ready && !flight);
            const unsigned long long left = __ballot(has);
            if (no_input && left && (uint32_t)__popcll(left) < P.tail_lanes) {
                park(has, pm, sample_index, true);
                has = false; pm.phase = rt::PH_DEAD;
                break;
            }
        }
    }
#if defined(RT_SHADE_CLOCKS)
    if (blockIdx.x == 77 && threadIdx.x == 64)
        printf("shade wave (fresh %d): %llu steps, %.1f lanes; clocks: take paths/camera %llu, pick kind %llu, book %llu, event %llu, primitives %llu, root boxes %llu, step end %llu, park + staging %llu\n",
               (int)FRESH, stick_steps, (double)stick_lanes / (double)(stick_steps ? stick_steps : 1), stick[0], stick[1], stick[2], stick[3], stick[4], stick[5], stick[6], stick[7]);
#endif
    /* unused tail of the last reserved chunk: mark the slots empty */
    for (uint32_t slot = out_cur + lane; slot < out_end && slot < P.out_capacity; slot += 64)
        *rt::ray_part(P.out_list, slot) = 0u;
    if (lane == 0 && parked_here) atomicAdd(&P.q->parked, parked_here);
    flush_counters<STATS>(cn, P.stats);
}

struct TreeParams {
    rt_dev_scene S;
    uint32_t *list;                  /* parked records: words 0..8 and 15 read, words 8..14 written when a triangle is accepted */
    uint32_t capacity;               /* slots of the list */
    uint32_t fixed_slots;            /* not 0: answer exactly this many slots (closest-hit probes); 0: what the round's record says */
    uint32_t finish_below;           /* the round's records are left to the finishing launch unless there are more than this */
    QueueCounters *q;                /* the record of the round that parked the list */
    uint64_t *stats;
    uint32_t *spill;                 /* stack entries beyond the LDS levels: (stack_cap - stack_levels) x threads of the launch */
    uint32_t stack_levels;           /* stack entries per ray kept in LDS */
    uint64_t *depth_hist;            /* counting builds: rays by the deepest stack they needed (RT_DEPTH_BINS bins), or NULL */
};

#define RT_DEPTH_BINS 32u
#ifndef RT_TREE_WAVES
#define RT_TREE_WAVES 7               /* waves per SIMD the tree kernel is compiled for (0: compiler's choice): 72 registers.  The kernel waits for
                                       * its dependent fetches, so resident waves are what it runs on: with the visited node's boxes staged in LDS
                                       * instead of twelve registers (RT_TREE_STAGE_BOXES) seven fit without spilling inside the loop (8 would
                                       * spill a dozen registers per lane: measured a quarter slower) */
#endif
#ifndef RT_TREE_LATE_FETCH
#define RT_TREE_LATE_FETCH 0          /* experiment: no record a step ahead at all (16 registers less: 7 waves too); as fast as 6 waves with the
                                       * record in registers, slower than 7 waves with the boxes staged */
#endif
#ifndef RT_TREE_STACK_LDS
#define RT_TREE_STACK_LDS 6u          /* stack entries per ray the tree kernel keeps in LDS (deeper ones go to device memory): with the staged
                                       * boxes (12 KB) and the leaf areas (4.5 KB) six levels (6 KB) let seven workgroups share a CU's LDS;
                                       * RT_GPU_TREE_STACK overrides it per scene handle */
#endif
#ifndef RT_TREE_LEAF_MIN
#define RT_TREE_LEAF_MIN 16           /* rays waiting at a leaf that make a leaf pass worth running beside node steps */
#endif
#ifndef RT_TREE_TOP
#define RT_TREE_TOP 0                 /* node records of the tree tops kept in LDS per workgroup (shared by up to two meshes); 0: none */
#endif
#ifndef RT_LEAF_MARKS
#define RT_LEAF_MARKS 1               /* leaf passes find the owner of a pair from a byte table of first pairs (0: binary search by lane exchange) */
#endif
#ifndef RT_LEAF_REDO
#define RT_LEAF_REDO 0                /* the ray that accepted a triangle computes its barycentrics again (0: the pair that won leaves them in LDS) */
#endif
#define RT_LEAF_WORDS_PER_WAVE (128u + (RT_LEAF_REDO ? 0u : 128u) + (RT_LEAF_MARKS ? 32u : 0u))   /* LDS words per wave for the leaf passes: 64 keys (2 words), 1024 mark bits, 64 x (beta, gamma) */
#ifndef RT_TREE_STAGE_BOXES
#define RT_TREE_STAGE_BOXES 1         /* 1: the boxes of the node a lane visits next (48 of the record's 64 bytes) wait in LDS, fetched there a step
                                       * ahead without passing through registers (global_load_lds); the two entries come through registers */
#endif
#if RT_TREE_TOP && RT_TREE_LATE_FETCH
#error "RT_TREE_TOP is not implemented for RT_TREE_LATE_FETCH"
#endif
#define RT_TREE_STAGE_LDS (RT_TREE_STAGE_BOXES ? (size_t)RT_BLOCK * 48u : (size_t)0)
#define RT_TREE_EXTRA_LDS ((size_t)(RT_BLOCK / 64) * RT_LEAF_WORDS_PER_WAVE * 4u + (size_t)RT_TREE_TOP * 64u + RT_TREE_STAGE_LDS)   /* bytes per workgroup for the leaf passes and the tree tops */
#ifndef RT_FINISH_SHADE_WEIGHT
#define RT_FINISH_SHADE_WEIGHT 8      /* finish kernel: weight (in 1/16ths) of a lane that wants a shade step in the vote for the kind of step to run (a shade step is long: it should run fuller; measured 2 ... 16) */
#endif
#ifndef RT_FINISH_REFILL
#define RT_FINISH_REFILL 16       /* finish kernel: free lanes take new records once this many are free */
#endif
#ifndef RT_FINISH_SPARSE
#define RT_FINISH_SPARSE 12       /* finish kernel: a wave with at most this many paths runs every kind of step in each iteration */
#endif
#define RT_FINISH_EXTRA_LDS ((size_t)(RT_BLOCK / 64) * RT_LEAF_WORDS_PER_WAVE * 4u)   /* the finish kernel's leaf passes */
#if RT_TREE_WAVES
#define RT_TREE_ATTR __attribute__((amdgpu_waves_per_eu(RT_TREE_WAVES, RT_TREE_WAVES)))
#else
#define RT_TREE_ATTR
#endif

/* ---- a leaf pass: the pending leaf triangles of all rays of the wave, 64 (ray, triangle) pairs per step ----
 * want_tri lanes hold a ray with an open leaf (tv.tri_cur < tv.tri_end); every lane of the wave must call.  Tests
 * the leaf's remaining triangles against the ray exactly as one query_tri_step per triangle would (src/main.c:449-482),
 * leaves the result in h and closes the leaf.  Returns whether a triangle was accepted (an any-hit ray is then done).
 * leaf_best: 64 keys, leaf_marks: 32 words, leaf_bary: 128 floats of the wave's own LDS. */
template <bool STATS>
__device__ __forceinline__ bool wave_leaf_pass(const rt_dev_scene &S, const bool want_tri, const unsigned long long tri_mask,
                                               const bool any_hit, rt::Hit &h, rt::Trav &tv, const uint32_t lane,
                                               unsigned long long *leaf_best,
                                               uint32_t *leaf_marks, float *leaf_bary, rt::Counters &cn)
{
    (void)leaf_marks; (void)leaf_bary;
    const int n_tri = __popcll(tri_mask);
    bool accepted = false;
    /* pairs are numbered ray by ray: ray of lane l owns pairs first..first+count-1 */
    const uint32_t count = want_tri ? tv.tri_end - tv.tri_cur : 0u;
    /* running sum over the 64 lanes by lane-shifted adds inside the vector unit (rows of 16, then the row
     * totals carried on): six instructions and no trip through the LDS crossbar, whose latency a scan by
     * lane exchange pays six times in a row */
    uint32_t incl = count;
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x111, 0xF, 0xF, false);    /* row_shr:1 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x112, 0xF, 0xF, false);    /* row_shr:2 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x114, 0xF, 0xF, false);    /* row_shr:4 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x118, 0xF, 0xF, false);    /* row_shr:8 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x142, 0xA, 0xF, false);    /* row_bcast:15 into rows 1 and 3 */
    incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, 0x143, 0xC, 0xF, false);    /* row_bcast:31 into rows 2 and 3 */
    const uint32_t first = incl - count;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    /* owners packed to the low lanes in lane order (= order of their first pairs): packed owner i is
     * the i-th ray with pairs */
    const uint32_t dest = want_tri ? rt::bits_below_lane(tri_mask)
                                   : (uint32_t)n_tri + rt::bits_below_lane(~tri_mask);
    const uint32_t packed_lane = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest * 4u), (int)lane);
#if RT_LEAF_MARKS
    /* which ray owns pair k: the rays mark their first pairs in a bit table (at most 64 x 16 pairs); the 64 bits
     * of a step's pairs number the owners: no search */
    if (lane < 32u) leaf_marks[lane] = 0u;
    if (want_tri) __hip_atomic_fetch_or(leaf_marks + (first >> 5), 1u << (first & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    uint32_t owners_before = 0;
#else
    const uint32_t packed_first = (uint32_t)__builtin_amdgcn_ds_permute((int)(dest * 4u), want_tri ? (int)first : -1);
#endif
    unsigned long long *best = leaf_best + lane;     /* this lane's slot as an owner */
    /* closest-hit rays: key = bits of t << 32 | (0xFFFFFFFE - place): smallest t, ties to the later triangle.
     * any-hit rays stop at the first accepted triangle: key = place << 32 | bits of t.  0 = walk one by one. */
    if (want_tri) *best = (h.t != h.t) ? 0ull
                        : any_hit ? ~0ull : (((unsigned long long)__float_as_uint(h.t) << 32) | 0xFFFFFFFFull);
    const uint32_t tri_first = (tv.tri_base + tv.tri_cur) | (any_hit ? 0x80000000u : 0u);
    for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t k = base + lane;
#if RT_LEAF_MARKS
        const uint32_t starts_lo = leaf_marks[base >> 5], starts_hi = leaf_marks[(base >> 5) + 1u];   /* the same words for every lane */
        const unsigned long long starts = ((unsigned long long)starts_hi << 32) | starts_lo;
        /* owners whose first pair lies at or before this lane's pair: those below the lane, and its own bit */
        const uint32_t pos = owners_before + rt::bits_below_lane(starts) + (uint32_t)((starts >> lane) & 1ull) - 1u;
        owners_before += (uint32_t)__popc(starts_lo) + (uint32_t)__popc(starts_hi);
#else
        uint32_t pos = 0;                            /* last packed owner whose first pair is <= k */
        for (uint32_t step = 32u; step; step >>= 1) {
            const uint32_t probe = pos + step;
            const uint32_t v = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(probe * 4u), (int)packed_first);
            if (v <= k) pos = probe;
        }
#endif
        const uint32_t owner = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(pos * 4u), (int)packed_lane);
        const int from = (int)(owner * 4u);
        const uint32_t own_first = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)first);
        const uint32_t own_word = (uint32_t)__builtin_amdgcn_ds_bpermute(from, (int)tri_first);
        const uint32_t own_tri = own_word & 0x7FFFFFFFu;
        const bool own_any = (own_word >> 31) != 0u;
        rt::f3 mo, mu;
        mo.x = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tv.mo.x)));
        mo.y = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tv.mo.y)));
        mo.z = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tv.mo.z)));
        mu.x = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tv.mu.x)));
        mu.y = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tv.mu.y)));
        mu.z = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(tv.mu.z)));
        const float entry_t = __int_as_float(__builtin_amdgcn_ds_bpermute(from, __float_as_int(h.t)));
        if (STATS) { if (lane == 0) cn.tri_steps++; if (k < total) cn.tri_lanes++; }
        if (k < total) {
            const uint32_t place = k - own_first;    /* position in the leaf's remaining triangles */
            const rt::u32x4 *q = reinterpret_cast<const rt::u32x4 *>(S.tris + own_tri + place);
            const rt::u32x4 fa = q[0], fb = q[1], fc = q[2];
            float t, beta, gamma;
            if (STATS && !own_any) cn.tri_tests++;   /* an any-hit ray counts its tests when it reads its result */
            const int verdict = rt::leaf_pair(fa, fb, fc, mo, mu, entry_t, t, beta, gamma);
            if (verdict != rt::PAIR_REJECTED) {
                unsigned long long *slot_best = leaf_best + owner;
                const unsigned long long key = verdict == rt::PAIR_UNORDERED ? 0ull
                    : own_any ? (((unsigned long long)place << 32) | (unsigned long long)__float_as_uint(t))
                              : (((unsigned long long)__float_as_uint(t) << 32) | (unsigned long long)(0xFFFFFFFEu - place));
#if RT_LEAF_REDO
                __hip_atomic_fetch_min(slot_best, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   /* result unused: no wait */
#else
                atomicMin(slot_best, key);
                /* a minimum with all-ones changes nothing and returns the slot as an LDS operation, in order behind
                 * the wave's updates above */
                if (key && atomicMin(slot_best, ~0ull) == key) {                 /* best so far: leave the barycentrics */
                    leaf_bary[2 * owner] = beta; leaf_bary[2 * owner + 1] = gamma;
                }
#endif
            }
        }
    }
    if (want_tri) {
        const unsigned long long key = atomicMin(best, ~0ull);
        uint32_t place = 0;
        if (key == 0ull) {                           /* unordered distance somewhere: the host's one-by-one walk */
            while (tv.tri_cur < tv.tri_end) {
                const int r = rt::query_tri_step<STATS>(S, any_hit, h, tv, cn);
                if (STATS && !any_hit) cn.tri_tests--;   /* counted by the pass above */
                if (r == rt::STEP_QUERY_DONE) { accepted = true; break; }   /* only an any-hit ray needs to know */
            }
        } else if (any_hit) {
            const uint32_t count_here = tv.tri_end - tv.tri_cur;
            if (key != ~0ull) {
                place = (uint32_t)(key >> 32);
                h.type = rt::HIT_TRIANGLE; h.t = __uint_as_float((uint32_t)key);
                h.id = tv.tri_cur + place; h.inst = (uint32_t)tv.inst;
                h.material = tv.material;
                accepted = true;
                if (STATS) cn.tri_tests += place + 1u;   /* the host stops at this triangle */
            } else if (STATS) cn.tri_tests += count_here;
        } else if ((uint32_t)key != 0xFFFFFFFFu) {
            place = 0xFFFFFFFEu - (uint32_t)key;
            h.type = rt::HIT_TRIANGLE; h.t = __uint_as_float((uint32_t)(key >> 32));
            h.id = tv.tri_cur + place; h.inst = (uint32_t)tv.inst;
            h.material = tv.material;
            accepted = true;
        }
        if (accepted && key != 0ull) {
#if RT_LEAF_REDO
            /* the winner's barycentrics: the same operations on the same operands again give the same bits
             * (cheaper than a second pass of LDS traffic for every pair, since few leaf visits end in a hit) */
            const rt::u32x4 *q = reinterpret_cast<const rt::u32x4 *>(S.tris + tv.tri_base + tv.tri_cur + place);
            float t2;
            (void)rt::leaf_pair(q[0], q[1], q[2], tv.mo, tv.mu, 3.402823466e+38f, t2, h.beta, h.gamma);
#else
            h.beta = leaf_bary[2 * lane]; h.gamma = leaf_bary[2 * lane + 1];
#endif
        }
        tv.tri_cur = tv.tri_end;
    }
    return accepted;
}

/* Answers the parked queries of one round: every record whose query is not complete has its ray walked through
 * the trees of the instances from the one it was parked for on (src/main.c:372-483).  A lane keeps only what the
 * walk itself needs (slot, closest distance, ray in mesh space, stack pointer, the prefetched node record): origin
 * and direction are read again from the record when the next instance's root box is due, and a triangle that is
 * accepted goes to the record at once (distance, identity, barycentrics: words 8..14), so an unanswered record is
 * simply left as it was.  Few registers and a short LDS stack are what let many waves share a SIMD here, and many
 * waves are what hides the dependent fetches of a walk. */
template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK) RT_TREE_ATTR
rt_tree_kernel(const TreeParams P)
{
    uint32_t slots = P.fixed_slots;
    if (!slots) {
        const uint32_t reserved = __builtin_amdgcn_readfirstlane(P.q->out_reserved);
        if (__builtin_amdgcn_readfirstlane(P.q->parked) <= P.finish_below) return;
        slots = reserved < P.capacity ? reserved : P.capacity;
    }
    if (blockIdx.x >= (slots + 255u) / 256u) return;
    extern __shared__ uint32_t lds[];
    rt::TreeStack stk;
    stk.base = lds + threadIdx.x; stk.stride = RT_BLOCK; stk.levels = (int)P.stack_levels;
    stk.spill = P.spill; stk.spill_stride = gridDim.x * RT_BLOCK; stk.thread = blockIdx.x * RT_BLOCK + threadIdx.x;
    rt::Counters cn; memset(&cn, 0, sizeof(cn));
    const uint32_t lane = threadIdx.x & 63;
    /* after the stacks: per wave 64 keys (8 bytes) and 64 barycentric pairs for the leaf passes */
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint32_t *leaf_area = lds + (size_t)RT_BLOCK * P.stack_levels + wave * RT_LEAF_WORDS_PER_WAVE;
    unsigned long long *leaf_best = reinterpret_cast<unsigned long long *>(leaf_area);
    uint32_t *leaf_marks = leaf_area + 128;              /* 32 words (if RT_LEAF_MARKS) */
    float *leaf_bary = reinterpret_cast<float *>(leaf_area + 128 + (RT_LEAF_MARKS ? 32 : 0));   /* 128 words (unless RT_LEAF_REDO) */
    (void)leaf_bary; (void)leaf_marks;

#if RT_TREE_TOP
    /* The first records of a mesh's node range are the top levels of its tree (breadth-first order): every walk
     * passes through them.  Each workgroup keeps a copy in LDS, so those visits cost an LDS read instead of a
     * 64-byte line through the vector memory pipeline. */
    uint32_t *tops = lds + (size_t)RT_BLOCK * P.stack_levels + (RT_BLOCK / 64) * RT_LEAF_WORDS_PER_WAVE;
    const uint32_t top_meshes = P.S.mesh_count < 2 ? (uint32_t)P.S.mesh_count : 2u;
    const uint32_t top_each = top_meshes ? RT_TREE_TOP / top_meshes : 0u;
    for (uint32_t m = 0; m < top_meshes; m++) {
        const uint32_t n = P.S.meshes[m].node_count < top_each ? P.S.meshes[m].node_count : top_each;
        const rt::u32x4 *src = reinterpret_cast<const rt::u32x4 *>(P.S.nodes + P.S.meshes[m].node_base);
        rt::u32x4 *dst = reinterpret_cast<rt::u32x4 *>(tops + (size_t)m * top_each * 16u);
        for (uint32_t i = threadIdx.x; i < n * 4u; i += RT_BLOCK) dst[i] = src[i];
    }
    __syncthreads();
#endif
    bool busy = false, any_hit = false;
    uint32_t slot = 0;
    rt::Hit h; h.type = rt::HIT_NONE; h.id = 0; h.inst = 0; h.t = 0; h.beta = 0; h.gamma = 0; h.material = -1; h.entered = false;
    rt::Trav tv; memset(&tv, 0, sizeof(tv)); tv.top = -1;
    uint32_t in_cur = 0, in_end = 0;
    bool input_done = false;
    uint32_t cur = RT_ENTRY_NONE;                        /* entry the lane visits next; its record is in ra..rd */
#if RT_TREE_LATE_FETCH
    const rt::u32x4 *node_at = nullptr;                  /* experiment: the record is fetched when the node step runs, not a step ahead (16 registers less) */
#elif RT_TREE_STAGE_BOXES
    rt::u32x4 rd = 0;
    /* per wave 3 quads x 64 lanes x 16 bytes, behind the leaf areas (and the tree tops): quad q of lane l at word (q * 64 + l) * 4 */
    uint32_t *node_stage = lds + (size_t)RT_BLOCK * P.stack_levels + (RT_BLOCK / 64) * RT_LEAF_WORDS_PER_WAVE + (size_t)RT_TREE_TOP * 16u + wave * (3u * 64u * 4u);
#else
    rt::u32x4 ra = 0, rb = 0, rc = 0, rd = 0;
#endif
    bool stepped = false;                                /* the lane needs its next record */
    bool scan_wait = false;                              /* the lane's instance is done (or its ray is new): it waits to test the next instance's root box */
    int depth_max = 0;                                   /* counting builds: deepest stack of the lane's ray so far */

#if defined(RT_TREE_CLOCKS)      /* experiment: where a wave's time goes (one wave of the launch prints its clock counts per section) */
    unsigned long long tick[5] = {0, 0, 0, 0, 0}, tick_prev = __builtin_readcyclecounter(), tick_iterations = 0;
#define RT_TICK(i) { const unsigned long long now = __builtin_readcyclecounter(); tick[i] += now - tick_prev; tick_prev = now; }
#else
#define RT_TICK(i)
#endif
    /* counting builds: the histogram of deepest stacks is kept per workgroup in LDS and added to the device's at the end
     * (one global atomic per ray on 32 addresses made the counting launch fifty times as long as the plain one) */
    __shared__ uint32_t depth_bins[STATS ? RT_DEPTH_BINS : 1u];
    if (STATS) {
        if (threadIdx.x < RT_DEPTH_BINS) depth_bins[threadIdx.x] = 0u;
        __syncthreads();
    }
    auto finish = [&]() {                                /* the query is answered: whatever was accepted is in the record already */
        if (STATS && P.depth_hist)
            atomicAdd(depth_bins + ((uint32_t)depth_max < RT_DEPTH_BINS ? (uint32_t)depth_max : RT_DEPTH_BINS - 1u), 1u);
        busy = false;
    };

    for (;;) {
        /* ---- free lanes take the next parked rays; wait until a quarter of the wave is free so
         * that loading a record and testing its root box does not run for one or two lanes ---- */
        const int n_free = __popcll(__ballot(!busy));
        while (!input_done && (n_free >= RT_TREE_REFILL || n_free == 64)) {
            const unsigned long long need = __ballot(!busy);
            if (!need) break;
            if (in_cur >= in_end) {
                uint32_t v = 0;
                if (lane == 0) v = atomicAdd(&P.q->tree_next, RT_TREE_IN_BATCH);
                in_cur = __builtin_amdgcn_readfirstlane(v);
                if (in_cur >= slots) { input_done = true; break; }
                in_end = in_cur + RT_TREE_IN_BATCH < slots ? in_cur + RT_TREE_IN_BATCH : slots;
            }
            const uint32_t rank = rt::bits_below_lane(need);
            const uint32_t idx = in_cur + rank;
            if (!busy && idx < in_end) {
                const uint32_t *w = rt::ray_part(P.list, idx);
                const uint32_t flags = w[0], t_bits = w[8], w15 = w[15];
                if ((flags & 0x81u) == 1u) {             /* a record, and its query is not complete */
                    slot = idx;
                    any_hit = ((flags >> 3) & 3u) == (uint32_t)rt::RAY_SHADOW;
                    h.t = __uint_as_float(t_bits);
                    tv.inst = (int)(w15 & 0xFFFFu) - 1;  /* walk_advance moves on to the instance the ray was parked for */
                    tv.top = -1; tv.tri_cur = 0; tv.tri_end = 0;
                    cur = RT_ENTRY_NONE;
                    depth_max = 0;
                    busy = true; scan_wait = true;       /* new rays arrive together: their first root-box test usually runs in this very iteration */
                    if (STATS) cn.root_tests--;          /* the shade kernel already counted this root test */
                }
            }
            in_cur += (uint32_t)__popcll(need);
            if (in_cur > in_end) in_cur = in_end;
            if (__ballot(busy) == ~0ull) break;
        }
        const unsigned long long working = __ballot(busy);
        if (!working) break;
        RT_TICK(0)

        /* ---- node steps by the lane that owns the ray; leaves by the whole wave ------------- */
        const bool want_tri = busy && !stepped && !scan_wait && tv.tri_cur < tv.tri_end;
        const bool want_node = busy && !stepped && !scan_wait && !want_tri;
        const unsigned long long tri_mask = __ballot(want_tri);
        const int n_tri = __popcll(tri_mask), n_node = __popcll(__ballot(want_node));
        const bool run_node = n_node > 0 && (n_node >= n_tri || n_node >= RT_TREE_BOTH);
        const bool run_leaf = n_tri > 0 && (n_tri >= RT_TREE_LEAF_MIN || !run_node);
        /* the leaf pass goes first: the node records requested at the end of the previous iteration
         * have its whole duration to arrive */
        if (run_leaf) {
            h.type = rt::HIT_NONE;                       /* what this pass accepts, if anything */
            const bool accepted = wave_leaf_pass<STATS>(P.S, want_tri, tri_mask, any_hit, h, tv, lane, leaf_best, leaf_marks, leaf_bary, cn);
            if (want_tri) {
                stepped = true;
                if (h.type == rt::HIT_TRIANGLE) {        /* the closest hit so far: into the record (a later, closer one overwrites it) */
                    uint32_t *w = rt::ray_part(P.list, slot);
                    *reinterpret_cast<uint4 *>(w + 8) = make_uint4(__float_as_uint(h.t), (uint32_t)rt::HIT_TRIANGLE, h.id, h.inst);
                    *reinterpret_cast<uint2 *>(w + 12) = make_uint2(__float_as_uint(h.beta), __float_as_uint(h.gamma));
                    w[14] = (uint32_t)tv.material;
                }
                if (accepted && any_hit) { finish(); stepped = false; }
            }
        }
        RT_TICK(1)
        if (run_node) {
#if RT_TREE_STAGE_BOXES
            __builtin_amdgcn_s_waitcnt(0x0F70);          /* vmcnt(0): the boxes requested at the end of the previous iteration are in LDS */
#endif
            if (STATS) { if (lane == 0) cn.pop_steps++; if (want_node) cn.pop_lanes++; }
            if (want_node) {
#if RT_TREE_LATE_FETCH
                const rt::u32x4 ra = node_at[0], rb = node_at[1], rc = node_at[2], rd = node_at[3];
#elif RT_TREE_STAGE_BOXES
                const rt::u32x4 *staged = reinterpret_cast<const rt::u32x4 *>(node_stage) + lane;
                rt::u32x4 ra, rb, rc;
#if RT_TREE_TOP
                if (tv.mesh < top_meshes && cur < top_each) {        /* a record of the tree top: from the workgroup's copy */
                    const rt::u32x4 *t = reinterpret_cast<const rt::u32x4 *>(tops + ((size_t)tv.mesh * top_each + cur) * 16u);
                    ra = t[0]; rb = t[1]; rc = t[2]; rd = t[3];
                } else
#endif
                { ra = staged[0]; rb = staged[64]; rc = staged[128]; }
#endif
                rt::walk_node<STATS>(ra, rb, rc, rd, h, tv, stk, cur, cn); stepped = true;
                if (STATS && tv.top + 1 > depth_max) depth_max = tv.top + 1;
            }
        }
        RT_TICK(2)
        /* lanes whose instance is done gather before the next instance's root box is tested: the test runs when enough
         * of them wait, or when the wave has little else to do */
        bool scan_now = false;
        {
            const unsigned long long waiting = __ballot(scan_wait);
            if (waiting) {
                if (P.S.instance_count <= 1) scan_now = true;       /* only new rays ever wait, and they come together */
                else {
                    const int n_wait = __popcll(waiting), n_other = __popcll(__ballot(busy && !scan_wait));
                    scan_now = n_wait >= RT_TREE_SCAN_MIN || n_other < RT_TREE_SCAN_MIN;
                }
            }
        }
        /* ---- the one place records are requested: every lane that took a step (or a new ray) ---- */
        if (stepped || (scan_now && scan_wait)) {
            const bool may_scan = scan_wait;             /* a lane that has waited tests now; one that just finished its instance waits */
            stepped = false; scan_wait = false;
            const rt::u32x4 *q = nullptr;
            rt::f3 o = rt::mk(0, 0, 0), d = rt::mk(0, 0, 0);
            if (may_scan) {                              /* origin and direction are not kept in registers between root-box tests */
                const uint4 *r = reinterpret_cast<const uint4 *>(rt::ray_part(P.list, slot));
                const uint4 a = r[0], b = r[1];
                o = rt::mk(__uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x));
                d = rt::mk(__uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
            }
            const int next = rt::walk_advance<STATS>(P.S, o, d, h, tv, stk, cur, q, cn, may_scan);
            if (next == rt::NEXT_SCAN) scan_wait = true;
            if (next == rt::NEXT_NODE) {
#if RT_TREE_TOP && RT_TREE_STAGE_BOXES && !RT_TREE_LATE_FETCH
                if (tv.mesh < top_meshes && cur < top_each) { }      /* nothing to request: the node step reads the workgroup's copy */
                else
#elif RT_TREE_TOP
                if (tv.mesh < top_meshes && cur < top_each) {
                    const rt::u32x4 *t = reinterpret_cast<const rt::u32x4 *>(tops + ((size_t)tv.mesh * top_each + cur) * 16u);
                    ra = t[0]; rb = t[1]; rc = t[2]; rd = t[3];
                } else
#endif
#if RT_TREE_LATE_FETCH
                node_at = q;
#elif RT_TREE_STAGE_BOXES
                {
                    const uint32_t *g = reinterpret_cast<const uint32_t *>(q);
                    __builtin_amdgcn_global_load_lds(g, node_stage, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(g + 4, node_stage + 256, 16, 0, 0);
                    __builtin_amdgcn_global_load_lds(g + 8, node_stage + 512, 16, 0, 0);
                    rd = q[3];
                }
#else
                { ra = q[0]; rb = q[1]; rc = q[2]; rd = q[3]; }
#endif
            }
            else if (next == rt::NEXT_DONE) finish();
        }
        RT_TICK(3)
#if defined(RT_TREE_CLOCKS)
        tick_iterations++;
#endif
    }
    if (STATS && P.depth_hist) {
        __syncthreads();
        if (threadIdx.x < RT_DEPTH_BINS && depth_bins[threadIdx.x])
            atomicAdd(reinterpret_cast<unsigned long long *>(P.depth_hist + threadIdx.x), (unsigned long long)depth_bins[threadIdx.x]);
    }
#if defined(RT_TREE_CLOCKS)
    if (blockIdx.x == 77 && threadIdx.x == 64)
        printf("tree wave: %llu iterations; clocks: refill %llu, leaf pass %llu, node step %llu, advance+request %llu\n",
               tick_iterations, tick[0], tick[1], tick[2], tick[3]);
#endif
    flush_counters<STATS>(cn, P.stats);
}

/* ---- the last few paths of a launch: parked records walked to their end in place -------
 * Rounds of shade and tree launches stop paying when a handful of paths is left (a launch has a
 * floor of some 50 us and ends with its slowest ray).  This kernel takes the parked, not yet
 * answered records and runs each path to its end with the in-wave scheduler: shade, node and
 * triangle steps by ballot, trees walked by the lane itself. */
#if defined(RT_FINISH_CLOCKS)    /* experiment: where the longest-running wave of the finishing launch spends its time */
__device__ unsigned long long rt_finish_clock_max;
#endif
struct FinishParams {
    rt_dev_scene S;
    const uint32_t *lists[2];        /* round r parked into lists[r & 1] */
    uint32_t capacity;               /* slots of either list */
    QueueCounters *rounds;           /* the batch's records, 0 ... last_round */
    uint32_t last_round;             /* rounds the host enqueued after the first launch */
    uint32_t finish_below;
    uint64_t *totals;                /* [0] += shade launches that had work, [1] += paths lost to a full list */
    float *samples; uint64_t *stats;
    uint32_t seed0, seed1;
    int max_segments;
    float half_w, half_h;
};

template <bool STATS, bool SPECULAR>
__global__ void __launch_bounds__(RT_BLOCK)
rt_finish_kernel(const FinishParams P)
{
    /* The rounds ended with the first one that parked no more than finish_below records (the launches
     * of the later rounds returned at once), or with the last one enqueued: its records, if any, are
     * walked to their end here. */
    uint32_t round = 0, lost = 0;
    for (;; round++) {
        lost += __builtin_amdgcn_readfirstlane(P.rounds[round].overflow);
        if (round == P.last_round || __builtin_amdgcn_readfirstlane(P.rounds[round].parked) <= P.finish_below) break;
    }
    QueueCounters *q = P.rounds + round;
    const uint32_t reserved = __builtin_amdgcn_readfirstlane(q->out_reserved);
    const uint32_t slots = reserved < P.capacity ? reserved : P.capacity;
    if (blockIdx.x == 0 && threadIdx.x == 0 && P.totals && P.rounds[0].pad[0]) {
        atomicAdd(reinterpret_cast<unsigned long long *>(P.totals), (unsigned long long)(round + 1u));
        if (lost) atomicAdd(reinterpret_cast<unsigned long long *>(P.totals + 1), (unsigned long long)lost);
    }
    if (__builtin_amdgcn_readfirstlane(q->parked) == 0u || blockIdx.x >= (slots + 255u) / 256u) return;
    const uint32_t *list = P.lists[round & 1u];
    extern __shared__ uint32_t lds[];
    rt::Stack stk; stk.base = lds + threadIdx.x; stk.stride = RT_BLOCK;
    rt::Counters cn; memset(&cn, 0, sizeof(cn));
    const uint32_t lane = threadIdx.x & 63;
    rt::PathMachine<rt::CounterRng, STATS, SPECULAR, true> pm(P.S, stk, cn);   /* box tests with the ray's shared reciprocals where they are exact */
    pm.half_w = P.half_w; pm.half_h = P.half_h; pm.max_segments = P.max_segments;
    pm.rng.seed0 = P.seed0; pm.rng.seed1 = P.seed1;
    uint32_t sample_index = 0;
    uint32_t in_cur = 0, in_end = 0;
    bool input_done = false;
    pm.phase = rt::PH_DEAD;
    /* after the stacks: the wave's area for the leaf passes, as in the tree kernel */
    uint32_t *leaf_area = lds + (size_t)RT_BLOCK * P.S.stack_cap + (threadIdx.x >> 6) * RT_LEAF_WORDS_PER_WAVE;
    unsigned long long *leaf_best = reinterpret_cast<unsigned long long *>(leaf_area);
    uint32_t *leaf_marks = leaf_area + 128;
    float *leaf_bary = reinterpret_cast<float *>(leaf_area + 128 + (RT_LEAF_MARKS ? 32 : 0));
    auto on_end = [&]() {                                /* a path ended */
        const rt::f3 c = pm.path_value();
        float *dst = P.samples + 3 * (size_t)sample_index;
        dst[0] = c.x; dst[1] = c.y; dst[2] = c.z;
        pm.phase = rt::PH_DEAD;
    };
#if defined(RT_FINISH_CLOCKS)
    unsigned long long ftick[4] = {0, 0, 0, 0}, fcount[4] = {0, 0, 0, 0}, flanes[4] = {0, 0, 0, 0}, ftick_prev = __builtin_readcyclecounter();
#define RT_FTICK(i, lanes) { const unsigned long long now = __builtin_readcyclecounter(); ftick[i] += now - ftick_prev; ftick_prev = now; fcount[i]++; flanes[i] += (unsigned long long)(lanes); }
#else
#define RT_FTICK(i, lanes)
#endif
    /* The in-wave scheduler of PathMachine::run with one change: an open leaf is tested by the whole wave at once
     * (this kernel's waves run with few paths, and its duration is the longest path's chain of steps: a leaf of
     * sixteen triangles is one step of that chain instead of sixteen). */
    for (;;) {
        /* free lanes take the next records, a quarter wave at a time */
        while (!input_done) {
            const unsigned long long need = __ballot(pm.phase == rt::PH_DEAD);
            if (__popcll(need) < RT_FINISH_REFILL) break;
            if (in_cur >= in_end) {
                uint32_t v = 0;
                if (lane == 0) v = atomicAdd(&q->in_next, 64u);
                in_cur = __builtin_amdgcn_readfirstlane(v);
                if (in_cur >= slots) { input_done = true; break; }
                in_end = in_cur + 64u < slots ? in_cur + 64u : slots;
            }
            const uint32_t idx = in_cur + rt::bits_below_lane(need);
            if (pm.phase == rt::PH_DEAD && idx < in_end &&
                pm.load_record(rt::ray_part(list, idx), rt::path_part(list, P.capacity, idx), sample_index)) {
                if (!(rt::ray_part(list, idx)[0] & 0x80u)) {   /* the query still has a tree to walk */
                    pm.tv.inst = (int)(rt::ray_part(list, idx)[15] & 0xFFFFu);   /* instance the ray was parked for */
                    pm.phase = rt::PH_SCAN;
                    if (STATS) cn.root_tests--;          /* counted when the ray was parked */
                }
            }
            in_cur += (uint32_t)__popcll(need);
            if (in_cur > in_end) in_cur = in_end;
        }
        const int n_shade = __popcll(__ballot(pm.wants_shade()));
        const int n_tri = __popcll(__ballot(pm.wants_tri()));
        const int n_pop = __popcll(__ballot(pm.wants_pop()));
        if (!(n_shade | n_tri | n_pop)) break;
        RT_FTICK(0, n_shade + n_tri + n_pop)
        /* a full wave runs the kind of step most lanes want (a shade step is long, so it should run fuller);
         * a wave with few paths left runs every kind in turn, so that each path moves on in every iteration
         * (pop, then the leaf it opened, then the shading of the answer) */
        const bool sparse = n_shade + n_tri + n_pop <= RT_FINISH_SPARSE;
        const int s_shade = n_shade * RT_FINISH_SHADE_WEIGHT;    /* weights in 1/16ths */
        const int s_tri = n_tri * 16, s_pop = n_pop * 16;
        const bool vote_shade = s_shade >= s_tri && s_shade >= s_pop;
        const bool vote_pop = !vote_shade && s_pop >= s_tri;
        if (sparse ? n_pop > 0 : vote_pop) {
            if (STATS) { if (lane == 0) cn.pop_steps++; if (pm.wants_pop()) cn.pop_lanes++; }
            if (pm.wants_pop()) pm.after_tree(rt::query_pop_step<STATS, true>(P.S, pm.h, pm.tv, stk, cn));
            RT_FTICK(1, n_pop)
        }
        if (sparse || !(vote_shade || vote_pop)) {
            const bool want_tri = pm.wants_tri();
            const unsigned long long tri_mask = __ballot(want_tri);
            if (tri_mask) {
                const bool any_hit = pm.ray == rt::RAY_SHADOW;
                const bool accepted = wave_leaf_pass<STATS>(P.S, want_tri, tri_mask, any_hit, pm.h, pm.tv, lane,
                                                            leaf_best, leaf_marks, leaf_bary, cn);
                if (want_tri && accepted && any_hit) pm.after_tree(rt::STEP_QUERY_DONE);
                RT_FTICK(2, __popcll(tri_mask))
            }
        }
        if (sparse ? __ballot(pm.wants_shade()) != 0ull : vote_shade) {
            if (STATS) { if (lane == 0) cn.shade_steps++; if (pm.wants_shade()) cn.shade_lanes++; }
#if defined(RT_FINISH_CLOCKS)
            const int shade_lanes_now = __popcll(__ballot(pm.wants_shade()));
#endif
            pm.shade_step(on_end);
            RT_FTICK(3, shade_lanes_now)
        }
    }
#if defined(RT_FINISH_CLOCKS)
    if (lane == 0) {
        const unsigned long long all = ftick[0] + ftick[1] + ftick[2] + ftick[3];
        if (atomicMax(&rt_finish_clock_max, all) < all)
            printf("finish wave %u.%u: clocks %llu; iterations %llu (%.1f paths each); refill+ballots %llu; pop steps %llu: %llu clocks, %.1f lanes; "
                   "leaf passes %llu: %llu clocks, %.1f lanes; shade steps %llu: %llu clocks, %.1f lanes\n",
                   blockIdx.x, threadIdx.x >> 6, all, fcount[0], (double)flanes[0] / (double)(fcount[0] ? fcount[0] : 1), ftick[0],
                   fcount[1], ftick[1], (double)flanes[1] / (double)(fcount[1] ? fcount[1] : 1),
                   fcount[2], ftick[2], (double)flanes[2] / (double)(fcount[2] ? fcount[2] : 1),
                   fcount[3], ftick[3], (double)flanes[3] / (double)(fcount[3] ? fcount[3] : 1));
    }
#endif
    flush_counters<STATS>(cn, P.stats);
}

/* per pixel: running sum += its samples of this launch, in sample order; the
 * last launch of a call applies the host's final factor (src/main.c:788) */
struct ResolveParams {
    const float *samples; float *pixels;
    uint32_t x0, y0, w, h;
    uint32_t blocks_x, block_rows, band_blocks, band_rank, band_count;
    uint32_t out_pitch, compact;
    uint32_t launch_spp;
    uint32_t start_from_pixels;      /* 1: the running sum starts from the stored pixel */
    uint32_t finish;                 /* 1: multiply by norm */
    float norm;
};

__global__ void __launch_bounds__(RT_BLOCK)
rt_resolve_kernel(const ResolveParams P)
{
    const uint32_t wave = (blockIdx.x * RT_BLOCK + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    if (wave >= P.blocks_x * P.block_rows) return;
    uint32_t bx = wave % P.blocks_x, own_row = wave / P.blocks_x, by = own_row;
    if (P.band_blocks) {
        uint32_t k = own_row / P.band_blocks, in = own_row % P.band_blocks;
        by = (k * P.band_count + P.band_rank) * P.band_blocks + in;
    }
    const uint32_t lx = bx * 8 + (lane & 7), ly = by * 8 + (lane >> 3);
    if (!(lx < P.w && ly < P.h)) return;
    const size_t out = P.compact ? ((size_t)(own_row * 8 + (lane >> 3)) * P.out_pitch + lx)
                                 : ((size_t)(P.y0 + ly) * P.out_pitch + (P.x0 + lx));
    float *dst = P.pixels + 3 * out;
    rt::f3 total = rt::mk(0, 0, 0);
    if (P.start_from_pixels) total = rt::mk(dst[0], dst[1], dst[2]);
    const float *src = P.samples + 3 * ((size_t)wave * 64 + lane);
    const size_t sample_stride = 3 * (size_t)P.blocks_x * P.block_rows * 64;
    for (uint32_t i = 0; i < P.launch_spp; i++, src += sample_stride)
        total = rt::add(total, rt::mk(src[0], src[1], src[2]));
    if (P.finish) total = rt::scale(total, P.norm);
    dst[0] = total.x; dst[1] = total.y; dst[2] = total.z;
}

/* ---- replay of host jobs: one lane = one Job (src/main.h:65-72), walked in the
 * host's order with the job's own AES generator state ---------------------- */
struct DevJob { uint32_t x0, y0, w, h, spp; float norm; uint32_t serie[8]; };

struct JobParams {
    rt_dev_scene S;
    float *pixels; uint64_t *stats;
    const DevJob *jobs; const uint32_t *aes_table;
    uint32_t job_count, width, height;
    int max_segments;
    float half_w, half_h;
};

template <bool STATS>
__global__ void __launch_bounds__(RT_BLOCK)
rt_render_jobs_kernel(const JobParams P)
{
    extern __shared__ uint32_t lds[];
    uint32_t *td = lds;                         /* 256 words, then the stacks */
    for (int i = threadIdx.x; i < 256; i += RT_BLOCK) td[i] = P.aes_table[i];
    __syncthreads();
    rt::Stack stk; stk.base = lds + 256 + threadIdx.x; stk.stride = RT_BLOCK;
    rt::Counters cn; memset(&cn, 0, sizeof(cn));

    const uint32_t j = blockIdx.x * RT_BLOCK + threadIdx.x;
    bool live = j < P.job_count;
    DevJob job;
    memset(&job, 0, sizeof(job));
    if (live) job = P.jobs[j];
    live = live && job.w && job.h;

    rt::PathMachine<rt::AesRng, STATS> pm(P.S, stk, cn);
    pm.half_w = P.half_w; pm.half_h = P.half_h; pm.max_segments = P.max_segments;
    pm.rng.td = td;
    for (int k = 0; k < 4; k++) { pm.rng.a[k] = job.serie[k]; pm.rng.b[k] = job.serie[4 + k]; }
    pm.x = job.x0; pm.y = job.y0;
    rt::f3 total = rt::mk(0, 0, 0);
    uint32_t s = 0;
    pm.start_path();
    if (!live) pm.phase = rt::PH_DEAD;

    pm.run(RT_SHADE_WEIGHT, [&]() {
        total = rt::add(total, pm.path_value());
        s++;
        pm.start_path();
        if (s == job.spp) {                     /* pixel finished: store, move on in the host's order */
            total = rt::scale(total, job.norm);
            float *dst = P.pixels + 3 * ((size_t)P.width * pm.y + pm.x);
            dst[0] = total.x; dst[1] = total.y; dst[2] = total.z;
            total = rt::mk(0, 0, 0);
            s = 0;
            pm.x++;
            if (pm.x == job.x0 + job.w) { pm.x = job.x0; pm.y++; }
            if (pm.y == job.y0 + job.h) pm.phase = rt::PH_DEAD;
        }
    });
    flush_counters<STATS>(cn, P.stats);
}

/* ---- batched closest-hit probe, output in the host's Ray_Result layout ----- */
/* one Ray_Result (src/math.h:148-168 layout, 72 bytes) */
__device__ __forceinline__ void write_ray_result(uint8_t *dst, const rt::Hit &h, const rt::Surface &sf)
{
    uint32_t *w = reinterpret_cast<uint32_t *>(dst);
    float *f = reinterpret_cast<float *>(w);
    const bool hit = h.type != rt::HIT_NONE;
    w[0] = hit ? 1u : 0u;
    w[1] = hit ? (uint32_t)h.type : 0u;
    w[2] = h.id;
    w[3] = h.type == rt::HIT_TRIANGLE ? h.inst : 0u;
    f[4] = sf.p.x; f[5] = sf.p.y; f[6] = sf.p.z;
    f[7] = sf.n.x; f[8] = sf.n.y; f[9] = sf.n.z;
    f[10] = h.t; f[11] = sf.alpha; f[12] = h.beta; f[13] = h.gamma;
    f[14] = sf.u; f[15] = sf.v;
    w[16] = h.entered ? 1u : 0u;
    w[17] = (uint32_t)h.material;
}

struct HitParams {
    rt_dev_scene S;
    const float *rays;          /* 6 floats per ray */
    uint8_t *results;           /* 72 bytes per ray */
    uint64_t *stats;
    uint32_t count;
    float max_distance;
};

__global__ void __launch_bounds__(RT_BLOCK)
rt_find_hit_kernel(const HitParams P)
{
    extern __shared__ uint32_t lds[];
    rt::Stack stk; stk.base = lds + threadIdx.x; stk.stride = RT_BLOCK;
    rt::Counters cn; memset(&cn, 0, sizeof(cn));
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < P.count) {
        rt::f3 o = rt::ld3(P.rays + 6 * i), d = rt::ld3(P.rays + 6 * i + 3);
        rt::Hit h;
        rt::closest_hit<true>(P.S, o, d, P.max_distance, false, stk, h, cn);
        rt::Surface sf;
        rt::surface_of<true>(P.S, o, d, h, sf, cn);
        write_ray_result(P.results + 72 * (size_t)i, h, sf);
    }
    flush_counters<true>(cn, P.stats);
}

/* ---- closest-hit probes through the render kernels' own machinery: the ray's scene primitives
 * and root-box test as the shade kernel does them, a parked record for rt_tree_kernel when the
 * ray enters an instance, then the surface attributes.  Word 0 of the record: bit 0 = parked,
 * bit 6 = entered flag of the sphere tests; the tree kernel leaves it alone. */
struct HitPackParams {
    rt_dev_scene S;
    const float *rays; uint32_t *list; uint64_t *stats;
    uint32_t count; float max_distance;
};

__global__ void __launch_bounds__(RT_BLOCK)
rt_hit_pack_kernel(const HitPackParams P)
{
    rt::Counters cn; memset(&cn, 0, sizeof(cn));
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < P.count) {
        const rt::f3 o = rt::ld3(P.rays + 6 * i), d = rt::ld3(P.rays + 6 * i + 3);
        rt::Hit h;
        int park = -1;
        if (!rt::query_begin<true>(P.S, o, d, P.max_distance, false, h, cn)) park = rt::query_probe<true>(P.S, o, d, h, 0, cn);
        uint4 *q = reinterpret_cast<uint4 *>(rt::ray_part(P.list, i));
        const uint32_t flags = (park >= 0 ? 1u : 0u) | ((uint32_t)rt::RAY_SEGMENT << 3) | (h.entered ? 0x40u : 0u);
        q[0] = make_uint4(flags, 0u, __float_as_uint(o.x), __float_as_uint(o.y));
        q[1] = make_uint4(__float_as_uint(o.z), __float_as_uint(d.x), __float_as_uint(d.y), __float_as_uint(d.z));
        q[2] = make_uint4(__float_as_uint(h.t), (uint32_t)h.type, h.id, h.inst);
        q[3] = make_uint4(__float_as_uint(h.beta), __float_as_uint(h.gamma), (uint32_t)h.material, (uint32_t)(park >= 0 ? park : 0));
    }
    flush_counters<true>(cn, P.stats);
}

__global__ void __launch_bounds__(RT_BLOCK)
rt_hit_unpack_kernel(const HitPackParams P, uint8_t *results)
{
    rt::Counters cn; memset(&cn, 0, sizeof(cn));
    const uint32_t i = blockIdx.x * RT_BLOCK + threadIdx.x;
    if (i < P.count) {
        const uint4 *q = reinterpret_cast<const uint4 *>(rt::ray_part(P.list, i));
        const uint4 a = q[0], b = q[1], c = q[2], e = q[3];
        const rt::f3 o = rt::mk(__uint_as_float(a.z), __uint_as_float(a.w), __uint_as_float(b.x));
        const rt::f3 d = rt::mk(__uint_as_float(b.y), __uint_as_float(b.z), __uint_as_float(b.w));
        rt::Hit h;
        h.t = __uint_as_float(c.x); h.type = (int)c.y; h.id = c.z; h.inst = c.w;
        h.beta = __uint_as_float(e.x); h.gamma = __uint_as_float(e.y); h.material = (int)e.z;
        h.entered = (a.x & 0x40u) != 0u;
        rt::Surface sf;
        rt::surface_of<true>(P.S, o, d, h, sf, cn);
        write_ray_result(results + 72 * (size_t)i, h, sf);
    }
    flush_counters<true>(cn, P.stats);
}

/* zeroes `count` 32-bit words (the queue counters of the static work memory: as a kernel, so that inside a captured
 * graph the clearing is a kernel node ordered like every other launch of the sequence) */
__global__ void rt_clear_words_kernel(uint32_t *words, uint32_t count)
{
    for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) words[i] = 0u;
}

__global__ void rt_normalize_kernel(float *p, size_t n, float factor)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * factor;
}

/* display transfer of save_image (src/main.c:73-82) on device pixels: 3 bytes per pixel */
__global__ void rt_quantize_kernel(const float *p, size_t n, uint8_t *rgb)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rgb[i] = rt_gamma_byte(p[i]);
}

__global__ void rt_math_kernel(const float *x, const float *y, size_t n, float *s, float *c, float *l,
                               float *q, float *r, float *qc)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float sn, cs;
    rt_sincosf(x[i], &sn, &cs);
    if (s) s[i] = sn;
    if (c) c[i] = cs;
    if (l) l[i] = rt_logf(x[i]);
    if (q && y) q[i] = x[i] / y[i];
    if (qc && y) qc[i] = (rt::tame(y[i]) && rt::tame_or_zero(x[i])) ? rt::div_by_cached(x[i], y[i], rt::refined_rcp(y[i])) : x[i] / y[i];
    if (r) r[i] = rt::sqrt_rn(x[i]);
}

/* v[3i..3i+2] -> its normalisation and its length as the path code computes them (rt::unit) */
__global__ void rt_unit_kernel(const float *v, size_t n, float *u, float *len)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float length;
    const rt::f3 r = rt::unit(rt::mk(v[3 * i], v[3 * i + 1], v[3 * i + 2]), length);
    u[3 * i] = r.x; u[3 * i + 1] = r.y; u[3 * i + 2] = r.z;
    len[i] = length;
}

/* ------------------------------------------------------------------------ */
/* host side                                                                 */
/* ------------------------------------------------------------------------ */

static thread_local std::string g_error;

static int fail(int code, const std::string &what)
{
    g_error = what;
    return code;
}

#define RT_HIP(call)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (call);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(RT_GPU_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

/* ---- work memory of rt_gpu_render_region_static: one array per GPU, part of the code object ----
 * Layout (byte offsets): the two queue-counter records (256 each), the per-sample values
 * (12 bytes per path), then one parked list (160 bytes per slot).  `rt_static_scratch` is only a
 * name on the host: the array's address on a device comes from hipGetSymbolAddress, once per device
 * (static_scratch_of), and that pointer is what kernels and memsets get. */
#define RT_STATIC_MAX_PATHS 65536u                       /* (8x8 pixel blocks of the region) x 64 x samples per pixel */
#define RT_STATIC_LIST_SLOTS (RT_STATIC_MAX_PATHS + 8192u)   /* + one partly used 64-slot chunk for each of at most 128 waves */
#define RT_STATIC_SAMPLES_AT 512u
#define RT_STATIC_LIST_AT (RT_STATIC_SAMPLES_AT + RT_STATIC_MAX_PATHS * 12u)
#define RT_STATIC_BYTES (RT_STATIC_LIST_AT + (size_t)RT_STATIC_LIST_SLOTS * RT_RECORD_WORDS * 4u)
static __device__ __align__(256) unsigned char rt_static_scratch[RT_STATIC_BYTES];
static_assert(RT_STATIC_LIST_AT % 256 == 0, "the parked list starts on a 256-byte boundary");

#define RT_MAX_DEVICES 64
static std::mutex g_static_mutex;
static unsigned char *g_static_ptr[RT_MAX_DEVICES];     /* device address of rt_static_scratch on device i (NULL: not asked yet) */

/* address of this device's copy of the array; the device must be current */
static int static_scratch_of(int device, unsigned char **out)
{
    *out = nullptr;
    if (device < 0 || device >= RT_MAX_DEVICES) return fail(RT_GPU_ERR_ARGUMENT, "device index above 63");
    std::lock_guard<std::mutex> guard(g_static_mutex);
    if (!g_static_ptr[device]) {
        void *p = nullptr;
        hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(rt_static_scratch));
        if (e != hipSuccess || !p)
            return fail(RT_GPU_ERR_DEVICE, std::string("hipGetSymbolAddress(rt_static_scratch): ") + hipGetErrorString(e));
        g_static_ptr[device] = static_cast<unsigned char *>(p);
    }
    *out = g_static_ptr[device];
    return 0;
}

#define RT_KINDS 4                    /* detailed counters are kept per kind of launch: */
enum { KIND_FRESH = 0, KIND_RESUME = 1, KIND_TREE = 2, KIND_FINISH = 3 };
#define RT_MAX_LANES 8
#define RT_MAX_PASSES 8               /* passes over a batch whose parked lists are smaller than the batch */
#define RT_EXTRA_PASS_ROUNDS 6u       /* rounds enqueued for the second and later passes */

/* Work memory of one batch of samples in flight: a frame's batches go round-robin over the lanes,
 * each on its own stream, so the latency-bound last rounds of one batch run beside the
 * machine-filling first launch of the next. */
struct BatchLane {
    hipStream_t stream = nullptr;
    float *d_samples = nullptr; uint64_t sample_bytes = 0;     /* per-sample values between shading and the per-pixel sum */
    uint32_t *d_parked[2] = {nullptr, nullptr}; uint64_t parked_records = 0;   /* paths in transit, ping-pong */
    QueueCounters *d_rounds = nullptr;                         /* one record per round of each pass */
    uint32_t *d_spill = nullptr;                               /* the tree launches' stack entries beyond the LDS levels (made with the lane's other work memory) */
    hipEvent_t done = nullptr;                                 /* after the lane's latest per-pixel sum */
};

struct Knobs {                        /* environment, read once per scene handle */
    uint64_t scratch_mb = 4096, parked_mb = 98304;
    uint32_t lanes = 3, max_rounds = 12, tail_lanes = 32;
    uint32_t parked_percent = 40;      /* size of the lists of paths in transit a handle starts with, in percent of what every path of a batch would need */
    uint64_t batch_paths = 80u << 20;
    uint32_t tree_stack = RT_TREE_STACK_LDS;   /* stack entries per ray the tree kernel keeps in LDS */
    int64_t finish_below = -1;         /* < 0: by the size of the batch */
    bool trace_rounds = false;
};

struct rt_gpu_scene {
    int device = 0;
    rt_dev_scene dev;
    float field_of_view = 0.0f;
    std::vector<void *> allocations;
    uint64_t *d_stats = nullptr;      /* RT_KINDS x RT_STAT_WORDS counters, then [rounds that had work, paths lost] */
    uint32_t *d_aes_table = nullptr;
    float *d_frame = nullptr; size_t frame_floats = 0;
    QueueCounters *d_queue = nullptr; /* closest-hit probes */
    BatchLane lanes[RT_MAX_LANES];
    hipEvent_t ev_start = nullptr;
    hipStream_t host_stream = nullptr;                        /* rt_gpu_render_region (host image) orders its render here */
    Knobs knobs;
    uint32_t shade_blocks = 0, resume_blocks = 0, tree_blocks = 0, finish_blocks = 0;   /* persistent grid sizes */
    uint32_t shade_blocks_shared = 0;                         /* the launch that starts paths while other lanes of the handle are busy */
    uint32_t tree_levels = 0;                                 /* stack entries per ray in the tree kernel's LDS (the rest in a lane's d_spill) */
    uint64_t spill_bytes = 0;                                 /* size of a lane's d_spill */
    uint64_t *d_depth_hist = nullptr;                         /* RT_DEPTH_BINS counters: rays by deepest stack (launches with detailed counters) */
    bool specular = true;                                     /* some material is a mirror or glass */
    struct Timer { hipEvent_t first, second; int kind; };
    std::vector<Timer> timers;                                /* pending kernel timings */
    std::vector<hipEvent_t> spare_events;
    double kind_ms[RT_KINDS] = {0.0, 0.0, 0.0, 0.0};
    uint32_t launches = 0, passes = 0;
    uint64_t samples = 0;
    uint32_t *more_passes = nullptr, *d_more_passes = nullptr;   /* one word of pinned host memory and its device address: a batch needed a second pass */
    bool lists_full = false;                                  /* the lanes' lists are sized for every path of a batch from now on */
    double create_ms[2] = {0.0, 0.0};                         /* flatten, allocate + upload */
    uint64_t resident_bytes = 0;
};

template <class T>
static int upload(rt_gpu_scene *s, const std::vector<T> &host, const T **dev)
{
    *dev = nullptr;
    size_t bytes = sizeof(T) * (host.empty() ? 1 : host.size());
    void *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return fail(RT_GPU_ERR_MEMORY, "hipMalloc failed while uploading the scene");
    s->allocations.push_back(p);
    if (!host.empty()) RT_HIP(hipMemcpy(p, host.data(), sizeof(T) * host.size(), hipMemcpyHostToDevice));
    *dev = static_cast<const T *>(p);
    return 0;
}

/* combined InvSubBytes + InvMixColumns column table of the AES decryption
 * round (FIPS-197 5.3): word x = {14,9,13,11} . Sinv[x], lowest byte = row 0 */
static void build_aes_table(uint32_t table[256])
{
    auto xt = [](uint8_t a) { return (uint8_t)((a << 1) ^ ((a & 0x80) ? 0x1b : 0)); };
    auto mulg = [&](uint8_t a, uint8_t b) {
        uint8_t p = 0;
        while (b) { if (b & 1) p ^= a; a = xt(a); b >>= 1; }
        return p;
    };
    uint8_t sbox[256], inv[256];
    for (int x = 0; x < 256; x++) {
        uint8_t i = 0;
        if (x) for (int y = 1; y < 256; y++) if (mulg((uint8_t)x, (uint8_t)y) == 1) { i = (uint8_t)y; break; }
        uint8_t s = i, r = i;
        for (int k = 0; k < 4; k++) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        sbox[x] = s ^ 0x63;
    }
    for (int x = 0; x < 256; x++) inv[sbox[x]] = (uint8_t)x;
    for (int x = 0; x < 256; x++) {
        uint8_t s = inv[x];
        table[x] = (uint32_t)mulg(s, 14) | ((uint32_t)mulg(s, 9) << 8) | ((uint32_t)mulg(s, 13) << 16) |
                   ((uint32_t)mulg(s, 11) << 24);
    }
}

/* the shade kernel instantiation for (new paths?, detailed counters?, scene has mirror or glass?) */
template <bool FRESH, bool STATS>
static void launch_shade_for(bool specular, uint32_t grid, hipStream_t stream, const ShadeParams &P)
{
    const size_t lds = FRESH ? RT_POOL_LDS : RT_STAGE_LDS;
    if (specular) hipLaunchKernelGGL((rt_shade_kernel<FRESH, STATS, true>), dim3(grid), dim3(RT_BLOCK), lds, stream, P);
    else hipLaunchKernelGGL((rt_shade_kernel<FRESH, STATS, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, P);
}
static void launch_shade(bool fresh, bool detailed, bool specular, uint32_t grid, hipStream_t stream, const ShadeParams &P)
{
    if (fresh) { if (detailed) launch_shade_for<true, true>(specular, grid, stream, P); else launch_shade_for<true, false>(specular, grid, stream, P); }
    else { if (detailed) launch_shade_for<false, true>(specular, grid, stream, P); else launch_shade_for<false, false>(specular, grid, stream, P); }
}

extern "C" {

int rt_gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

const char *rt_gpu_last_error(void) { return g_error.c_str(); }

/* for the library's other translation units (not part of the public header) */
int rt_gpu_set_error(int code, const char *what) { return fail(code, what ? what : ""); }

static void free_lane(BatchLane &L)
{
    if (L.d_samples) (void)hipFree(L.d_samples);
    for (int k = 0; k < 2; k++) if (L.d_parked[k]) (void)hipFree(L.d_parked[k]);
    if (L.d_spill) (void)hipFree(L.d_spill);
    L.d_spill = nullptr;
    L.d_samples = nullptr; L.sample_bytes = 0;
    L.d_parked[0] = L.d_parked[1] = nullptr; L.parked_records = 0;
}

void rt_gpu_scene_destroy(rt_gpu_scene *s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    (void)hipDeviceSynchronize();
    for (void *p : s->allocations) (void)hipFree(p);
    if (s->d_frame) (void)hipFree(s->d_frame);
    for (auto &L : s->lanes) {
        free_lane(L);
        if (L.d_rounds) (void)hipFree(L.d_rounds);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.stream) (void)hipStreamDestroy(L.stream);
    }
    if (s->more_passes) (void)hipHostFree(s->more_passes);
    if (s->ev_start) (void)hipEventDestroy(s->ev_start);
    if (s->host_stream) (void)hipStreamDestroy(s->host_stream);
    for (auto &t : s->timers) { (void)hipEventDestroy(t.first); (void)hipEventDestroy(t.second); }
    for (auto e : s->spare_events) (void)hipEventDestroy(e);
    delete s;
}

/* RT_GPU_* environment knobs, validated once when a scene handle is made */
static int read_knobs(Knobs &k)
{
    auto number = [](const char *name, uint64_t lo, uint64_t hi, uint64_t &out) -> int {
        const char *e = getenv(name);
        if (!e || !*e) return 0;
        char *end = nullptr;
        unsigned long long v = strtoull(e, &end, 10);
        if (*end || v < lo || v > hi)
            return fail(RT_GPU_ERR_ARGUMENT, std::string(name) + "=" + e + " is outside " + std::to_string(lo) + " ... " + std::to_string(hi));
        out = v;
        return 0;
    };
    uint64_t v;
    int rc = 0;
    if ((rc = number("RT_GPU_SCRATCH_MB", 1, 1u << 20, k.scratch_mb))) return rc;
    if ((rc = number("RT_GPU_PARKED_MB", 1, 1u << 20, k.parked_mb))) return rc;
    v = k.lanes; if ((rc = number("RT_GPU_STREAMS", 1, RT_MAX_LANES, v))) return rc; k.lanes = (uint32_t)v;
    v = k.max_rounds; if ((rc = number("RT_GPU_MAX_ROUNDS", 0, 255, v))) return rc; k.max_rounds = (uint32_t)v;
    v = k.parked_percent; if ((rc = number("RT_GPU_PARKED_PERCENT", 1, 100, v))) return rc; k.parked_percent = (uint32_t)v;
    v = k.tail_lanes; if ((rc = number("RT_GPU_TAIL_LANES", 0, 64, v))) return rc; k.tail_lanes = (uint32_t)v;
    if ((rc = number("RT_GPU_BATCH_PATHS", 64, 1ull << 40, k.batch_paths))) return rc;
    v = k.tree_stack; if ((rc = number("RT_GPU_TREE_STACK", 1, 64, v))) return rc; k.tree_stack = (uint32_t)v;
    v = 0; if (getenv("RT_GPU_FINISH_BELOW")) { if ((rc = number("RT_GPU_FINISH_BELOW", 0, 0xFFFFFFFFull, v))) return rc; k.finish_below = (int64_t)v; }
    k.trace_rounds = getenv("RT_GPU_TRACE_ROUNDS") != nullptr;
    return 0;
}

int rt_gpu_scene_create(const rt_scene *scene, rt_gpu_scene **out)
{
    if (!scene || !out) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_scene_create: NULL argument");
    *out = nullptr;
    if (rt_gpu_device_count() <= 0) return fail(RT_GPU_ERR_NO_DEVICE, "no HIP device is visible");
    const auto t_begin = std::chrono::steady_clock::now();
    rt_flat_scene flat;
    if (rt_flatten_scene(scene, flat) != 0) return fail(RT_GPU_ERR_SCENE, "scene rejected: " + flat.error);
    const auto t_flat = std::chrono::steady_clock::now();

    /* a parked path record keeps the instance to walk and the light being sampled in 16 bits each */
    if (flat.instances.size() > 0xFFFEu) return fail(RT_GPU_ERR_SCENE, "scene rejected: more than 65534 mesh instances");
    if (flat.lights.size() > 0xFFFFu) return fail(RT_GPU_ERR_SCENE, "scene rejected: more than 65535 lights");

    rt_gpu_scene *s = new rt_gpu_scene();
    int rc = 0;
    do {
        if (hipGetDevice(&s->device) != hipSuccess) { rc = fail(RT_GPU_ERR_DEVICE, "hipGetDevice failed"); break; }
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, s->device) != hipSuccess) { rc = fail(RT_GPU_ERR_DEVICE, "hipGetDeviceProperties failed"); break; }
        size_t lds_need = (size_t)RT_BLOCK * flat.stack_cap * 4 + RT_TREE_EXTRA_LDS + 64;
        if (lds_need > prop.sharedMemPerBlock) {
            rc = fail(RT_GPU_ERR_SCENE, "box tree too deep for the LDS traversal stack (" + std::to_string(flat.stack_cap) + " levels)");
            break;
        }
        memset(&s->dev, 0, sizeof(s->dev));
        s->dev.cam = flat.camera;
        s->field_of_view = flat.field_of_view;
        s->dev.light_count = (int32_t)flat.lights.size();
        s->dev.sphere_count = (int32_t)flat.spheres.size();
        s->dev.plane_count = (int32_t)flat.planes.size();
        s->dev.instance_count = (int32_t)flat.instances.size();
        s->dev.material_count = (int32_t)flat.materials.size();
        {   /* does any primitive use a mirror or glass material?  (the table may list materials nothing uses) */
            auto is_specular = [&](int32_t id) { return id >= 0 && (size_t)id < flat.materials.size() && flat.materials[(size_t)id].kind != 0u; };
            bool any = false;
            for (const auto &x : flat.spheres) any = any || is_specular(x.material);
            for (const auto &x : flat.planes) any = any || is_specular(x.material);
            for (const auto &x : flat.instances) any = any || is_specular(x.material);
            s->specular = any;
        }
        s->dev.texture_count = (int32_t)flat.textures.size();
        s->dev.mesh_count = (int32_t)flat.meshes.size();
        s->dev.stack_cap = flat.stack_cap;
#if defined(RT_STACK_CLAMP)
        if (s->dev.stack_cap > RT_STACK_CLAMP) s->dev.stack_cap = RT_STACK_CLAMP;
#endif
        if ((rc = upload(s, flat.lights, &s->dev.lights))) break;
        if ((rc = upload(s, flat.spheres, &s->dev.spheres))) break;
        if ((rc = upload(s, flat.planes, &s->dev.planes))) break;
        if ((rc = upload(s, flat.instances, &s->dev.instances))) break;
        if ((rc = upload(s, flat.meshes, &s->dev.meshes))) break;
        if ((rc = upload(s, flat.materials, &s->dev.materials))) break;
        if ((rc = upload(s, flat.textures, &s->dev.textures))) break;
        if ((rc = upload(s, flat.nodes, &s->dev.nodes))) break;
        if ((rc = upload(s, flat.tris, &s->dev.tris))) break;
        if ((rc = upload(s, flat.attrs, &s->dev.attrs))) break;
        if ((rc = upload(s, flat.texels, &s->dev.texels))) break;
        if ((rc = read_knobs(s->knobs))) break;
        std::vector<uint64_t> zeros(RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS + RT_DEPTH_BINS, 0);
        const uint64_t *st = nullptr;
        if ((rc = upload(s, zeros, &st))) break;
        s->d_stats = const_cast<uint64_t *>(st);
        s->d_depth_hist = s->d_stats + RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS;
        s->tree_levels = s->knobs.tree_stack < s->dev.stack_cap ? s->knobs.tree_stack : s->dev.stack_cap;
        std::vector<uint32_t> qzero(sizeof(QueueCounters) / 4, 0);
        const uint32_t *qp = nullptr;
        if ((rc = upload(s, qzero, &qp))) break;
        s->d_queue = reinterpret_cast<QueueCounters *>(const_cast<uint32_t *>(qp));
        {   /* persistent grids: as many workgroups as the device keeps resident */
            const size_t lds_bytes = (size_t)RT_BLOCK * s->dev.stack_cap * 4;
            int per_cu_shade = 0, per_cu_tree = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_shade, rt_shade_kernel<true, false, true>, RT_BLOCK, RT_POOL_LDS) != hipSuccess || per_cu_shade < 1) per_cu_shade = 1;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_tree, rt_tree_kernel<false>, RT_BLOCK, (size_t)RT_BLOCK * s->tree_levels * 4 + RT_TREE_EXTRA_LDS) != hipSuccess || per_cu_tree < 1) per_cu_tree = 1;
            int per_cu_finish = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_finish, rt_finish_kernel<false, true>, RT_BLOCK, lds_bytes + RT_FINISH_EXTRA_LDS) != hipSuccess || per_cu_finish < 1) per_cu_finish = 1;
            s->finish_blocks = (uint32_t)(prop.multiProcessorCount * per_cu_finish);
            int per_cu_resume = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_resume, rt_shade_kernel<false, false, true>, RT_BLOCK, RT_STAGE_LDS) != hipSuccess || per_cu_resume < 1) per_cu_resume = 1;
            s->resume_blocks = (uint32_t)(prop.multiProcessorCount * per_cu_resume);
            if (const char *e = getenv("RT_GPU_FRESH_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v < per_cu_shade) per_cu_shade = v; }
            if (const char *e = getenv("RT_GPU_RESUME_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v < per_cu_resume) per_cu_resume = v; }
            if (const char *e = getenv("RT_GPU_TREE_BLOCKS_PER_CU")) { int v = atoi(e); if (v >= 1 && v < per_cu_tree) per_cu_tree = v; }
            s->resume_blocks = (uint32_t)(prop.multiProcessorCount * per_cu_resume);
            s->tree_blocks = (uint32_t)(prop.multiProcessorCount * per_cu_tree);
            s->shade_blocks = (uint32_t)(prop.multiProcessorCount * per_cu_shade);
            /* With four frames in flight (calls on three other lanes of the handle still running) the launch that starts the
             * paths leaves a quarter of every CU to the other frames' tree and resumed-path launches: it is bound by the
             * vector unit and loses little with 3 of 4 workgroups, they wait for memory half of the time and fill the
             * gap (measured with four calls in flight: 21.9 -> 21.4 ms per frame; with three in flight or alone the full
             * grid is faster: 22.1 against 22.5 ms) */
            int per_cu_shared = per_cu_shade > 1 ? per_cu_shade - 1 : per_cu_shade;
            if (const char *e = getenv("RT_GPU_FRESH_BLOCKS_PER_CU_SHARED")) { int v = atoi(e); if (v >= 1 && v <= per_cu_shade) per_cu_shared = v; }
            s->shade_blocks_shared = (uint32_t)(prop.multiProcessorCount * per_cu_shared);
            s->tree_blocks = (uint32_t)(prop.multiProcessorCount * per_cu_tree);
            /* one slice per thread of a tree launch for the stack entries beyond the LDS levels (at least one entry: the
             * pointer is never null) */
            const uint32_t deep = s->dev.stack_cap > s->tree_levels ? s->dev.stack_cap - s->tree_levels : 1u;
            s->spill_bytes = (uint64_t)s->tree_blocks * RT_BLOCK * deep * 4u;
        }
        {   /* one stream, one record table and one event per batch lane; their work memory comes with the first render */
            bool ok = hipEventCreateWithFlags(&s->ev_start, hipEventDisableTiming) == hipSuccess &&
                      hipStreamCreateWithFlags(&s->host_stream, hipStreamNonBlocking) == hipSuccess;
            for (uint32_t i = 0; ok && i < s->knobs.lanes; i++) {
                BatchLane &L = s->lanes[i];
                ok = hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking) == hipSuccess &&
                     hipEventCreateWithFlags(&L.done, hipEventDisableTiming) == hipSuccess &&
                     hipMalloc(reinterpret_cast<void **>(&L.d_rounds), sizeof(QueueCounters) * (s->knobs.max_rounds + 1) * RT_MAX_PASSES) == hipSuccess;
            }
            if (!ok) { rc = fail(RT_GPU_ERR_DEVICE, "could not create the streams, events and round records of the batch lanes"); break; }
        }
        {   /* one word the device can write and the host can read without a copy: "a batch needed a second pass" */
            void *host_word = nullptr, *dev_word = nullptr;
            if (hipHostMalloc(&host_word, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dev_word, host_word, 0) != hipSuccess) {
                if (host_word) (void)hipHostFree(host_word);
                rc = fail(RT_GPU_ERR_DEVICE, "could not map a word of host memory for the device");
                break;
            }
            s->more_passes = static_cast<uint32_t *>(host_word); s->d_more_passes = static_cast<uint32_t *>(dev_word);
            *s->more_passes = 0u;
            s->lists_full = s->knobs.parked_percent >= 100;
        }
        {   /* this device's address of the static work memory (rt_gpu_render_region_static), asked once per device */
            unsigned char *unused = nullptr;
            if ((rc = static_scratch_of(s->device, &unused))) break;
        }
        std::vector<uint32_t> table(256);
        build_aes_table(table.data());
        const uint32_t *tb = nullptr;
        if ((rc = upload(s, table, &tb))) break;
        s->d_aes_table = const_cast<uint32_t *>(tb);
    } while (0);
    if (rc) { rt_gpu_scene_destroy(s); return rc; }
    (void)hipDeviceSynchronize();
    const auto t_end = std::chrono::steady_clock::now();
    s->create_ms[0] = std::chrono::duration<double, std::milli>(t_flat - t_begin).count();
    s->create_ms[1] = std::chrono::duration<double, std::milli>(t_end - t_flat).count();
    s->resident_bytes = sizeof(rt_dev_node) * flat.nodes.size() + sizeof(rt_dev_tri) * flat.tris.size() + sizeof(rt_dev_attr) * flat.attrs.size() +
                        sizeof(float) * flat.texels.size();
    *out = s;
    return RT_GPU_OK;
}

int rt_gpu_scene_info(const rt_gpu_scene *s, rt_gpu_scene_info_t *out)
{
    if (!s || !out) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_scene_info: NULL argument");
    memset(out, 0, sizeof(*out));
    out->flatten_ms = s->create_ms[0]; out->upload_ms = s->create_ms[1];
    out->resident_bytes = s->resident_bytes;
    out->mesh_count = (uint32_t)s->dev.mesh_count; out->instance_count = (uint32_t)s->dev.instance_count;
    out->texture_count = (uint32_t)s->dev.texture_count; out->stack_levels = s->dev.stack_cap;
    for (uint32_t i = 0; i < s->knobs.lanes; i++)
        out->work_bytes += s->lanes[i].sample_bytes + 2 * s->lanes[i].parked_records * RT_RECORD_WORDS * 4;
    return RT_GPU_OK;
}

int rt_gpu_scene_set_camera(rt_gpu_scene *s, const void *camera)
{
    if (!s || !camera) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_scene_set_camera: NULL argument");
    const rt_camera *c = static_cast<const rt_camera *>(camera);
    rt_flatten_camera(c, s->dev.cam);
    s->field_of_view = c->field_of_view;
    return RT_GPU_OK;
}

/* work memory of one lane: grows, never shrinks; false = the device has no room (everything of the lane freed) */
static bool ensure_lane(BatchLane &L, uint64_t sample_bytes, uint64_t parked_records, uint64_t spill_bytes)
{
    if (L.sample_bytes >= sample_bytes && L.parked_records >= parked_records) return true;
    (void)hipStreamSynchronize(L.stream);
    free_lane(L);
    bool ok = hipMalloc(reinterpret_cast<void **>(&L.d_samples), sample_bytes) == hipSuccess &&
              hipMalloc(reinterpret_cast<void **>(&L.d_spill), spill_bytes) == hipSuccess;
    for (int k = 0; ok && k < 2; k++)
        ok = hipMalloc(reinterpret_cast<void **>(&L.d_parked[k]), parked_records * RT_RECORD_WORDS * 4) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); free_lane(L); return false; }
    L.sample_bytes = sample_bytes; L.parked_records = parked_records;
    return true;
}

static hipEvent_t take_event(rt_gpu_scene *s)
{
    if (!s->spare_events.empty()) { hipEvent_t e = s->spare_events.back(); s->spare_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
}

static void settle_timers(rt_gpu_scene *s)
{
    for (auto &t : s->timers) {
        float ms = 0.0f;
        if (hipEventSynchronize(t.second) == hipSuccess && hipEventElapsedTime(&ms, t.first, t.second) == hipSuccess)
            s->kind_ms[t.kind] += ms;
        s->spare_events.push_back(t.first);
        s->spare_events.push_back(t.second);
    }
    s->timers.clear();
}

/* a launch with, if asked for, a pair of events around it on its stream (its duration lands in kind_ms[kind]) */
extern "C++" template <class Launch>
static int timed_launch(rt_gpu_scene *s, bool timed, int kind, hipStream_t stream, Launch &&launch)
{
    hipEvent_t e0 = timed ? take_event(s) : nullptr, e1 = timed ? take_event(s) : nullptr;
    if (timed && !(e0 && e1)) {                          /* no events to be had: run untimed */
        if (e0) s->spare_events.push_back(e0);
        if (e1) s->spare_events.push_back(e1);
        e0 = e1 = nullptr;
    }
    bool recorded = e0 && hipEventRecord(e0, stream) == hipSuccess;
    launch();
    hipError_t err = hipGetLastError();
    if (recorded && err == hipSuccess && hipEventRecord(e1, stream) == hipSuccess) s->timers.push_back({e0, e1, kind});
    else if (e0) { s->spare_events.push_back(e0); s->spare_events.push_back(e1); }
    if (err != hipSuccess) return fail(RT_GPU_ERR_DEVICE, std::string("kernel launch: ") + hipGetErrorString(err));
    return 0;
}

/* which 8-row block rows of the region this launch walks; *own_rows = pixel rows among them */
static int launch_geometry(const rt_gpu_options *o, rt_rect2u region, RenderParams &P, uint32_t *own_rows)
{
    uint32_t block_rows_total = (region.size.y + 7) / 8;
    P.blocks_x = (region.size.x + 7) / 8;
    P.band_blocks = 0; P.band_rank = 0; P.band_count = 1;
    P.block_rows = block_rows_total;
    if (o && o->band_rows) {
        if (o->band_rows % 8 || !o->band_count || o->band_rank >= o->band_count)
            return fail(RT_GPU_ERR_ARGUMENT, "band_rows must be a multiple of 8 and band_rank < band_count");
        P.band_blocks = o->band_rows / 8; P.band_rank = o->band_rank; P.band_count = o->band_count;
        uint32_t bands = (block_rows_total + P.band_blocks - 1) / P.band_blocks;
        uint32_t own = 0, rows = 0;
        for (uint32_t b = P.band_rank; b < bands; b += P.band_count) {
            uint32_t left = block_rows_total - b * P.band_blocks;
            own += left < P.band_blocks ? left : P.band_blocks;
            uint32_t left_rows = region.size.y - b * o->band_rows;
            rows += left_rows < o->band_rows ? left_rows : o->band_rows;
        }
        /* a short band can only be the last band of the region, hence the last one a rank owns:
         * the kernel's packed numbering k*band_blocks+i stays valid */
        P.block_rows = own;
        *own_rows = rows;
        return 0;
    }
    *own_rows = region.size.y;
    return 0;
}

static void launch_finish(bool detailed, bool specular, uint32_t grid, size_t lds, hipStream_t stream, const FinishParams &F)
{
    if (detailed) {
        if (specular) hipLaunchKernelGGL((rt_finish_kernel<true, true>), dim3(grid), dim3(RT_BLOCK), lds, stream, F);
        else hipLaunchKernelGGL((rt_finish_kernel<true, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, F);
    } else {
        if (specular) hipLaunchKernelGGL((rt_finish_kernel<false, true>), dim3(grid), dim3(RT_BLOCK), lds, stream, F);
        else hipLaunchKernelGGL((rt_finish_kernel<false, false>), dim3(grid), dim3(RT_BLOCK), lds, stream, F);
    }
}

int rt_gpu_render_region_device(rt_gpu_scene *s, float *device_pixels, uint32_t width, uint32_t height,
                                rt_rect2u region, uint32_t spp, uint64_t seed,
                                const rt_gpu_options *o, void *hip_stream)
{
    if (!s || !device_pixels) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_render_region_device: NULL argument");
    if (o && o->struct_size != sizeof(rt_gpu_options)) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_options.struct_size mismatch");
    if (!width || !height || !region.size.x || !region.size.y || !spp) return fail(RT_GPU_ERR_ARGUMENT, "empty image, region or sample count");
    if ((uint64_t)region.point.x + region.size.x > width || (uint64_t)region.point.y + region.size.y > height)
        return fail(RT_GPU_ERR_ARGUMENT, "region exceeds the image");
    if ((uint64_t)width * height > 0xFFFFFFFFull) return fail(RT_GPU_ERR_ARGUMENT, "image too large for 32-bit pixel indices");
    if (o && o->max_segments >= (1u << 24)) return fail(RT_GPU_ERR_ARGUMENT, "max_segments must be below 2^24");
    RT_HIP(hipSetDevice(s->device));
    hipStream_t caller = static_cast<hipStream_t>(hip_stream);

    RenderParams G;
    memset(&G, 0, sizeof(G));
    uint32_t own_rows = 0;
    int rc = launch_geometry(o, region, G, &own_rows);
    if (rc) return rc;
    if (G.block_rows == 0) return RT_GPU_OK;
    if (width > 0xFFFFu || height > 0xFFFFu) return fail(RT_GPU_ERR_ARGUMENT, "image sides above 65535 are not supported");
    const Knobs &K = s->knobs;
    const uint32_t flags = o ? o->flags : 0;
    const bool detailed = (flags & RT_GPU_FLAG_DETAILED_STATS) != 0;
    const bool timed = detailed || (flags & RT_GPU_FLAG_KERNEL_TIMES) != 0;
    const bool accumulate = (flags & RT_GPU_FLAG_ACCUMULATE) != 0;
    const uint32_t frames = (o && o->frames > 1) ? o->frames : 1;
    if ((uint64_t)spp * frames + (o ? o->sample_offset : 0) > 0xFFFFFFFFull) return fail(RT_GPU_ERR_ARGUMENT, "sample indices exceed 32 bits");
    const uint64_t frame_floats = (flags & RT_GPU_FLAG_COMPACT_BANDS) ? (uint64_t)own_rows * region.size.x * 3
                                                                      : (uint64_t)width * height * 3;
    /* packed bands: the default distance between frames is the packed buffer of the rank with the most rows, the same
     * for every rank (what an all-gather of the ranks' buffers needs), not this rank's own row count */
    const uint64_t default_stride = (flags & RT_GPU_FLAG_COMPACT_BANDS) && o && o->band_rows
        ? (uint64_t)rt_gpu_multi_rows_per_rank(region.size.y, o->band_rows, o->band_count) * region.size.x * 3 : frame_floats;
    const uint64_t frame_stride = (o && o->frame_stride) ? o->frame_stride : default_stride;
    if (frames > 1 && frame_stride < frame_floats) return fail(RT_GPU_ERR_ARGUMENT, "frame_stride is smaller than a frame");
    const uint64_t total_spp = (uint64_t)spp * frames;
    if (o && o->lane >= K.lanes) return fail(RT_GPU_ERR_ARGUMENT, "options.lane must be below RT_GPU_STREAMS (" + std::to_string(K.lanes) + ")");
    const uint32_t first_lane = o ? o->lane : 0;

    ShadeParams P;
    memset(&P, 0, sizeof(P));
    P.S = s->dev;
    P.S.cam.focal = rt_camera_focal(s->field_of_view, height);
    P.width = width; P.height = height;
    P.x0 = region.point.x; P.y0 = region.point.y; P.w = region.size.x; P.h = region.size.y;
    P.blocks_x = G.blocks_x; P.block_rows = G.block_rows;
    P.band_blocks = G.band_blocks; P.band_rank = G.band_rank; P.band_count = G.band_count;
    P.sample_offset = o ? o->sample_offset : 0;
    P.seed0 = (uint32_t)seed; P.seed1 = (uint32_t)(seed >> 32);
    P.max_segments = (o && o->max_segments) ? (int)o->max_segments : 64;
    P.half_w = (float)width / 2.0f; P.half_h = (float)height / 2.0f;

    ResolveParams R;
    memset(&R, 0, sizeof(R));
    R.pixels = device_pixels;
    R.x0 = P.x0; R.y0 = P.y0; R.w = P.w; R.h = P.h;
    R.blocks_x = G.blocks_x; R.block_rows = G.block_rows;
    R.band_blocks = G.band_blocks; R.band_rank = G.band_rank; R.band_count = G.band_count;
    R.compact = (flags & RT_GPU_FLAG_COMPACT_BANDS) ? 1 : 0;
    R.out_pitch = R.compact ? region.size.x : width;
    R.norm = rt_pixel_norm((o && o->spp_total) ? o->spp_total : spp);

    /* The call's samples are rendered in batches (whole samples per pixel of the region; whole frames when
     * the call makes several).  A batch needs 12 bytes per path for the sample values and two lists of
     * 160-byte records for the paths in transit between launches.  Lists that hold every path of the batch
     * (+ one partly used chunk per wave) cannot fill up; where the budgets or the device do not allow
     * that, the lists get what is allowed and the batch's paths start in several passes: the launch that
     * starts paths stops taking portions of new work when its list is nearly full, the rounds carry those
     * paths to their end, and the next pass goes on with the portions left (on the device: the host
     * enqueues the passes a batch can need at worst, and passes without work return at once).  Only when
     * that would take more than RT_MAX_PASSES passes the batches get smaller.  Batches go round-robin
     * over the lanes: the budgets are shared between the lanes in use. */
    const uint64_t units = (uint64_t)G.blocks_x * G.block_rows;
    const uint64_t items_per_sample = units * 64;
    const uint64_t record_bytes = (uint64_t)RT_RECORD_WORDS * 4;
    uint64_t batch_spp = (K.batch_paths + items_per_sample / 2) / items_per_sample;
    if (batch_spp < 1) batch_spp = 1;
    if (batch_spp > total_spp) batch_spp = total_spp;
    if (batch_spp * items_per_sample > 0xFFFFFF00ull / 3) batch_spp = (0xFFFFFF00ull / 3) / items_per_sample;
    if (batch_spp < 1) return fail(RT_GPU_ERR_ARGUMENT, "region too large for one launch");
    if (!s->lists_full && __atomic_load_n(s->more_passes, __ATOMIC_RELAXED)) s->lists_full = true;   /* an earlier batch of this handle needed a second pass */
    size_t free_bytes = 0, device_bytes = 0;
    if (hipMemGetInfo(&free_bytes, &device_bytes) != hipSuccess) { (void)hipGetLastError(); free_bytes = ~(size_t)0; }
    uint64_t held = 0;
    for (uint32_t i = 0; i < K.lanes; i++) held += s->lanes[i].sample_bytes + 2 * s->lanes[i].parked_records * record_bytes;
    /* are calls of this handle under way on at least three other lanes (four frames in flight)?  With fewer the full grid is faster */
    uint32_t busy_lanes = 0;
    for (uint32_t i = 0; i < K.lanes; i++)
        if (i != first_lane % K.lanes && s->lanes[i].done && hipEventQuery(s->lanes[i].done) == hipErrorNotReady) busy_lanes++;
    (void)hipGetLastError();
    const bool others_busy = busy_lanes >= 3;
    uint32_t shade_grid = 0, resume_grid = 0, lanes_used = 0, lane_cap = K.lanes, passes = 1;
    uint64_t batches = 0, capacity = 0, out_limit = 0;
    (void)device_bytes;
    for (;;) {
        if (frames > 1) {
            if (batch_spp < spp) {
                /* not even one frame fits a batch: the frames one after the other, each in its sample batches */
                rt_gpu_options one = *o;
                one.frames = 1; one.frame_stride = 0;
                for (uint32_t k = 0; k < frames; k++) {
                    one.sample_offset = o->sample_offset + k * spp;
                    rc = rt_gpu_render_region_device(s, device_pixels + (size_t)k * frame_stride, width, height, region, spp, seed, &one, hip_stream);
                    if (rc) return rc;
                }
                return RT_GPU_OK;
            }
            batch_spp -= batch_spp % spp;                /* whole frames per batch */
            batches = (total_spp + batch_spp - 1) / batch_spp;
        } else {
            batches = (total_spp + batch_spp - 1) / batch_spp;
            batch_spp = (total_spp + batches - 1) / batches;        /* batches of equal size */
        }
        lanes_used = (uint32_t)(batches < lane_cap ? batches : lane_cap);
        /* no more workgroups than there are portions of fresh work for their waves (small regions); the
         * resumed-path launches get no more workgroups than the first launch */
        const uint64_t portions_max = units * ((batch_spp + RT_PORTION_SPP - 1) / RT_PORTION_SPP);
        shade_grid = others_busy ? s->shade_blocks_shared : s->shade_blocks;
        if ((portions_max + 3) / 4 < shade_grid) shade_grid = (uint32_t)((portions_max + 3) / 4);
        resume_grid = s->resume_blocks < shade_grid ? s->resume_blocks : shade_grid;
        const uint64_t batch_paths = batch_spp * items_per_sample;
        const uint64_t shade_waves = (uint64_t)shade_grid * (RT_BLOCK / 64), resume_waves = (uint64_t)resume_grid * (RT_BLOCK / 64);
        /* every wave of a launch leaves at most one reserved chunk partly used */
        const uint64_t full = (batch_paths + shade_waves * RT_CHUNK + RT_CHUNK - 1) / RT_CHUNK * RT_CHUNK;
        const uint64_t sample_bytes = batch_paths * 12;
        /* the budgets are the handle's: a lane gets its share whether or not the other lanes are in use;
         * what the lanes already hold is freed before they grow, so it counts as available on the device */
        uint64_t lists_bytes = (K.parked_mb << 20) / lane_cap;
        bool ok = sample_bytes * lane_cap <= (K.scratch_mb << 20);
        const bool within_budget = ok;
        if (free_bytes != ~(size_t)0) {
            const double per_lane = 0.9 * ((double)free_bytes + (double)held) / (double)lanes_used - (double)sample_bytes;
            if (per_lane <= 0.0) { ok = false; lists_bytes = 0; }
            else if (per_lane < (double)lists_bytes) lists_bytes = (uint64_t)per_lane;
        }
        uint64_t allowed = lists_bytes / (2 * record_bytes) / RT_CHUNK * RT_CHUNK;       /* slots of either list */
        if (allowed > 0xFFFFFF00ull) allowed = 0xFFFFFF00ull;
        capacity = full; passes = 1; out_limit = 0;
        /* a wave learns how full the list is when it reserves a chunk of it.  Once the limit is passed a wave
         * can still fill the chunk it has, take one more portion of new paths, park all of them and the paths
         * its lanes and its pool hold (the first of these reservations tells it to stop), and leave a chunk
         * partly used; the resumed-path launches park no more records than they were given, plus a partly
         * used chunk per wave */
        const uint64_t after_look = shade_waves * (RT_CHUNK + 64u * RT_PORTION_SPP + 64u + RT_POOL + RT_CHUNK);
        const uint64_t margin = after_look + resume_waves * RT_CHUNK;
        const uint64_t padding = shade_waves * RT_CHUNK;             /* reserved slots that hold no record */
        if (!s->lists_full) {
            /* Lists for every path of the batch are the worst case (a scene whose every path walks a tree at once);
             * in the room scenes a quarter of the paths is in transit at most.  A handle therefore starts with lists
             * of parked_percent of that size, the passes being the safety net, and only a handle whose batch really
             * needed a second pass (the device says so in `more_passes`) gets full lists from its next call on.
             * Batches so small that the waves' margins would eat the saving keep full lists. */
            const uint64_t modest = (full * K.parked_percent / 100 + RT_CHUNK - 1) / RT_CHUNK * RT_CHUNK;
            if (modest >= 2 * (margin + 2 * padding) && modest < allowed) allowed = modest;
        }
        if (ok && full > allowed) {
            capacity = allowed;
            if (capacity > margin + 2 * padding) {
                out_limit = capacity - margin;
                const uint64_t started_per_pass = out_limit - padding;   /* a pass ends early only with this many paths parked */
                const uint64_t needed = (batch_paths + started_per_pass - 1) / started_per_pass;
                if (needed <= RT_MAX_PASSES) passes = (uint32_t)needed; else ok = false;
            } else ok = false;
        }
        for (uint32_t i = 0; ok && i < lanes_used; i++) ok = ensure_lane(s->lanes[(first_lane + i) % K.lanes], sample_bytes, capacity, s->spill_bytes);
        if (ok) break;
        if (batch_spp == 1 || (frames > 1 && batch_spp == spp)) {
            if (lane_cap > 1) { lane_cap--; continue; }              /* the smallest batches: fewer of them in flight */
            if (frames > 1) { batch_spp = spp - 1; continue; }       /* fall back to one frame after the other */
            if (!within_budget || (K.parked_mb << 20) / (2 * record_bytes) < full)
                return fail(RT_GPU_ERR_MEMORY, "RT_GPU_SCRATCH_MB / RT_GPU_PARKED_MB are too small for one sample per pixel of this region (" +
                            std::to_string(sample_bytes >> 20) + " + " + std::to_string((2 * full * record_bytes) >> 20) + " MiB would hold it)");
            return fail(RT_GPU_ERR_MEMORY, "the device has no room for the work memory of one sample per pixel of this region (" +
                        std::to_string((sample_bytes + 2 * full * record_bytes) >> 20) + " MiB)");
        }
        batch_spp = frames > 1 ? (batch_spp / spp + 1) / 2 * spp : (batch_spp + 1) / 2;      /* smaller batches and try again */
    }

    const size_t lds = (size_t)RT_BLOCK * s->dev.stack_cap * 4;
    const size_t tree_lds = (size_t)RT_BLOCK * s->tree_levels * 4 + RT_TREE_EXTRA_LDS;
    const uint32_t resolve_grid = (uint32_t)((units * 64 + RT_BLOCK - 1) / RT_BLOCK);
    /* parked records at or below which one launch finishes the paths.  Measured on three scenes: a frame of
     * 1024x1024x64 (67 M paths) is shortest when the rounds stop at about a million records (the survivors of
     * several rounds are long paths); a batch of half that or less when they stop at two million, which for an
     * eighth of the frame is right after the first launch (a round's two launches cost as much for two million
     * records as the finishing launch saves by getting fewer) */
    const uint64_t finish_paths = batch_spp * items_per_sample <= (1ull << 25) ? (2u << 20) : (1u << 20);
    const uint32_t finish_below = K.finish_below >= 0 ? (uint32_t)K.finish_below : (uint32_t)finish_paths;
    /* a wave that has no input left and fewer paths than this hands them to the next round; needs the
     * finishing launch to end the batch (without it the rounds run to the last path themselves) */
    P.tail_lanes = finish_below == 0 ? 0 : K.tail_lanes;
    P.finish_below = finish_below;
    P.out_capacity = (uint32_t)capacity;
    P.out_limit = (uint32_t)out_limit;
    const uint32_t rounds = K.max_rounds;

    RT_HIP(hipEventRecord(s->ev_start, caller));
    hipEvent_t previous_sum = nullptr;                   /* per-pixel sums run in batch order: a pixel's samples are added in sample order */
    for (uint64_t batch = 0, first64 = 0; first64 < total_spp; batch++, first64 += batch_spp) {
        BatchLane &L = s->lanes[(first_lane + batch % lanes_used) % K.lanes];     /* only the lanes whose work memory was made ready above */
        if (!L.d_samples || !L.d_parked[0] || !L.d_parked[1] || !L.d_rounds || !L.d_spill) return fail(RT_GPU_ERR_DEVICE, "internal: batch lane without work memory");
        hipStream_t stream = L.stream;
        if (batch < lanes_used) RT_HIP(hipStreamWaitEvent(stream, s->ev_start, 0));
        const uint32_t first = (uint32_t)first64;
        P.first_sample = first;
        P.launch_spp = (uint32_t)(total_spp - first64 < batch_spp ? total_spp - first64 : batch_spp);
        P.samples = L.d_samples;
        RT_HIP(hipMemsetAsync(L.d_rounds, 0, sizeof(QueueCounters) * (rounds + 1) * passes, stream));
        P.portion_next = &L.d_rounds[0].unit_next;       /* the passes of a batch share the cursor over its new paths */
        for (uint32_t pass = 0; pass < passes; pass++) {
            QueueCounters *Q = L.d_rounds + (size_t)pass * (rounds + 1);      /* the pass's round records */
            /* the second and later passes are the safety net of lists smaller than the batch: they get fewer rounds
             * (an unneeded launch costs 3.6 us; whatever their last round leaves is the finishing launch's) */
            const uint32_t pass_rounds = pass == 0 || rounds < RT_EXTRA_PASS_ROUNDS ? rounds : RT_EXTRA_PASS_ROUNDS;
            /* round 0: new paths */
            P.stats = s->d_stats + KIND_FRESH * RT_STAT_WORDS;
            P.q = Q; P.q_in = nullptr; P.in_list = nullptr; P.out_list = L.d_parked[0];
            P.chunk = RT_CHUNK;
            P.pass_count = s->d_stats + RT_KINDS * RT_STAT_WORDS + 2;
            P.more_passes = pass > 0 && !s->lists_full ? s->d_more_passes : nullptr;
            if ((rc = timed_launch(s, timed, KIND_FRESH, stream, [&]() { launch_shade(true, detailed, s->specular, shade_grid, stream, P); }))) return rc;
            /* rounds 1 ...: answer the parked queries, resume those paths.  What a round has to do stands in
             * the previous round's record; launches of rounds that are not needed return at once. */
            for (uint32_t r = 1; r <= pass_rounds; r++) {
                TreeParams T;
                memset(&T, 0, sizeof(T));
                T.S = P.S; T.list = L.d_parked[(r - 1) & 1]; T.capacity = P.out_capacity; T.finish_below = finish_below;
                T.q = Q + (r - 1); T.stats = s->d_stats + KIND_TREE * RT_STAT_WORDS;
                T.spill = L.d_spill; T.stack_levels = s->tree_levels; T.depth_hist = s->d_depth_hist;
                if ((rc = timed_launch(s, timed, KIND_TREE, stream, [&]() {
                        if (detailed) hipLaunchKernelGGL(rt_tree_kernel<true>, dim3(s->tree_blocks), dim3(RT_BLOCK), tree_lds, stream, T);
                        else hipLaunchKernelGGL(rt_tree_kernel<false>, dim3(s->tree_blocks), dim3(RT_BLOCK), tree_lds, stream, T);
                    }))) return rc;
                P.stats = s->d_stats + KIND_RESUME * RT_STAT_WORDS;
                P.q_in = Q + (r - 1); P.q = Q + r;
                P.in_list = L.d_parked[(r - 1) & 1]; P.out_list = L.d_parked[r & 1];
                if ((rc = timed_launch(s, timed, KIND_RESUME, stream, [&]() { launch_shade(false, detailed, s->specular, resume_grid, stream, P); }))) return rc;
            }
            /* whatever the rounds left parked: one launch walks those paths to their end */
            FinishParams F;
            memset(&F, 0, sizeof(F));
            F.S = P.S; F.lists[0] = L.d_parked[0]; F.lists[1] = L.d_parked[1]; F.capacity = P.out_capacity;
            F.rounds = Q; F.last_round = pass_rounds; F.finish_below = finish_below;
            F.totals = s->d_stats + RT_KINDS * RT_STAT_WORDS;
            F.samples = P.samples; F.stats = s->d_stats + KIND_FINISH * RT_STAT_WORDS;
            F.seed0 = P.seed0; F.seed1 = P.seed1; F.max_segments = P.max_segments; F.half_w = P.half_w; F.half_h = P.half_h;
            if ((rc = timed_launch(s, timed, KIND_FINISH, stream, [&]() { launch_finish(detailed, s->specular, s->finish_blocks, lds + RT_FINISH_EXTRA_LDS, stream, F); }))) return rc;
        }

        if (previous_sum) RT_HIP(hipStreamWaitEvent(stream, previous_sum, 0));
        R.samples = L.d_samples;
        if (frames == 1) {
            R.launch_spp = P.launch_spp;
            R.start_from_pixels = (accumulate || first > 0) ? 1 : 0;
            R.finish = (!accumulate && first + P.launch_spp >= spp) ? 1 : 0;
            hipLaunchKernelGGL(rt_resolve_kernel, dim3(resolve_grid), dim3(RT_BLOCK), 0, stream, R);
            RT_HIP(hipGetLastError());
        } else {
            /* the batch holds whole frames: frame j of it owns samples j*spp ... of the batch */
            for (uint32_t j = 0; j < P.launch_spp / spp; j++) {
                ResolveParams F2 = R;
                F2.samples = L.d_samples + 3 * (size_t)j * spp * items_per_sample;
                F2.pixels = device_pixels + (size_t)(first / spp + j) * frame_stride;
                F2.launch_spp = spp;
                F2.start_from_pixels = accumulate ? 1 : 0;
                F2.finish = accumulate ? 0 : 1;
                hipLaunchKernelGGL(rt_resolve_kernel, dim3(resolve_grid), dim3(RT_BLOCK), 0, stream, F2);
                RT_HIP(hipGetLastError());
            }
        }
        RT_HIP(hipEventRecord(L.done, stream));
        previous_sum = L.done;
        s->launches++; s->passes += passes;
        if (K.trace_rounds) {                            /* debugging aid: wait for the batch and print its rounds */
            std::vector<QueueCounters> q((size_t)(rounds + 1) * passes);
            RT_HIP(hipStreamSynchronize(stream));
            RT_HIP(hipMemcpy(q.data(), L.d_rounds, sizeof(QueueCounters) * q.size(), hipMemcpyDeviceToHost));
            for (uint32_t pass = 0; pass < passes; pass++)
                for (uint32_t r = 0; r <= (pass == 0 || rounds < RT_EXTRA_PASS_ROUNDS ? rounds : RT_EXTRA_PASS_ROUNDS); r++) {
                    const QueueCounters &c = q[(size_t)pass * (rounds + 1) + r];
                    fprintf(stderr, "batch %llu pass %u round %u: parked %u records in %u slots of %llu%s\n", (unsigned long long)batch, pass, r, c.parked,
                            c.out_reserved, (unsigned long long)capacity, c.parked <= finish_below ? " -> finishing launch" : "");
                    if (c.overflow) fprintf(stderr, "batch %llu pass %u round %u: %u records found the list full\n", (unsigned long long)batch, pass, r, c.overflow);
                    if (c.parked <= finish_below) break;
                }
        }
    }
    /* the caller's stream goes on when every lane has made its last per-pixel sum */
    const uint64_t batches_made = (total_spp + batch_spp - 1) / batch_spp;
    for (uint32_t i = 0; i < lanes_used && i < batches_made; i++) RT_HIP(hipStreamWaitEvent(caller, s->lanes[(first_lane + i) % K.lanes].done, 0));
    if (s->timers.size() > 8192) settle_timers(s);
    s->samples += total_spp * own_rows * region.size.x;
    return RT_GPU_OK;
}

/* The render of rt_gpu_render_region_device for callers that may not allocate: every buffer the
 * kernels need beyond the caller's pixels lives in rt_static_scratch, and the launch sequence is
 * fixed (new paths -> the in-wave finishing kernel for every parked path -> per-pixel sums): no
 * read-back, no events, no hipMalloc / hipFree.  The finishing kernel learns how many slots the
 * first launch reserved from the counter record in device memory. */
int rt_gpu_render_region_static(rt_gpu_scene *s, float *device_pixels, uint32_t width, uint32_t height,
                                rt_rect2u region, uint32_t spp, uint64_t seed,
                                const rt_gpu_options *o, void *hip_stream)
{
    if (!s || !device_pixels) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_render_region_static: NULL argument");
    if (o && o->struct_size != sizeof(rt_gpu_options)) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_options.struct_size mismatch");
    if (!width || !height || !region.size.x || !region.size.y || !spp) return fail(RT_GPU_ERR_ARGUMENT, "empty image, region or sample count");
    if ((uint64_t)region.point.x + region.size.x > width || (uint64_t)region.point.y + region.size.y > height)
        return fail(RT_GPU_ERR_ARGUMENT, "region exceeds the image");
    if (width > 0xFFFFu || height > 0xFFFFu) return fail(RT_GPU_ERR_ARGUMENT, "image sides above 65535 are not supported");
    if (o && o->frames > 1) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_render_region_static renders one frame per call");
    if ((uint64_t)spp + (o ? o->sample_offset : 0) > 0xFFFFFFFFull) return fail(RT_GPU_ERR_ARGUMENT, "sample indices exceed 32 bits");
    if (o && o->max_segments >= (1u << 24)) return fail(RT_GPU_ERR_ARGUMENT, "max_segments must be below 2^24");
    RenderParams G;
    memset(&G, 0, sizeof(G));
    uint32_t own_rows = 0;
    int rc = launch_geometry(o, region, G, &own_rows);
    if (rc) return rc;
    if (G.block_rows == 0) return RT_GPU_OK;
    const uint64_t units = (uint64_t)G.blocks_x * G.block_rows;
    const uint64_t paths = units * 64 * spp;
    if (paths > RT_STATIC_MAX_PATHS)
        return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_render_region_static: " + std::to_string(units) + " 8x8 pixel blocks x 64 x " +
                    std::to_string(spp) + " samples = " + std::to_string(paths) + " paths exceed the static work memory (" +
                    std::to_string(RT_STATIC_MAX_PATHS) + " paths, e.g. a 64x64 tile at 16 samples per pixel)");
    RT_HIP(hipSetDevice(s->device));
    unsigned char *base = nullptr;
    if ((rc = static_scratch_of(s->device, &base))) return rc;
    hipStream_t stream = static_cast<hipStream_t>(hip_stream);
    QueueCounters *q_shade = reinterpret_cast<QueueCounters *>(base);        /* the one round of this launch sequence */
    float *samples = reinterpret_cast<float *>(base + RT_STATIC_SAMPLES_AT);
    uint32_t *list = reinterpret_cast<uint32_t *>(base + RT_STATIC_LIST_AT);
    const uint32_t flags = o ? o->flags : 0;
    const bool detailed = (flags & RT_GPU_FLAG_DETAILED_STATS) != 0;
    const bool accumulate = (flags & RT_GPU_FLAG_ACCUMULATE) != 0;

    ShadeParams P;
    memset(&P, 0, sizeof(P));
    P.S = s->dev;
    P.S.cam.focal = rt_camera_focal(s->field_of_view, height);
    P.stats = s->d_stats + KIND_FRESH * RT_STAT_WORDS;
    P.q = q_shade; P.portion_next = &q_shade->unit_next;
    P.samples = samples;
    P.out_list = list; P.out_capacity = RT_STATIC_LIST_SLOTS;
    P.chunk = RT_SMALL_CHUNK;
    P.tail_lanes = 32;
    P.width = width; P.height = height;
    P.x0 = region.point.x; P.y0 = region.point.y; P.w = region.size.x; P.h = region.size.y;
    P.blocks_x = G.blocks_x; P.block_rows = G.block_rows;
    P.band_blocks = G.band_blocks; P.band_rank = G.band_rank; P.band_count = G.band_count;
    P.first_sample = 0; P.launch_spp = spp;
    P.sample_offset = o ? o->sample_offset : 0;
    P.seed0 = (uint32_t)seed; P.seed1 = (uint32_t)(seed >> 32);
    P.max_segments = (o && o->max_segments) ? (int)o->max_segments : 64;
    P.half_w = (float)width / 2.0f; P.half_h = (float)height / 2.0f;

    /* one wave per portion of new work at most, never more than 128 waves: the list has a 64-slot chunk of slack for each */
    const uint64_t portions = units * ((spp + RT_PORTION_SPP - 1) / RT_PORTION_SPP);
    uint32_t shade_grid = (uint32_t)((portions + 3) / 4);
    if (shade_grid > 32u) shade_grid = 32u;
    if (shade_grid > s->shade_blocks) shade_grid = s->shade_blocks;

    hipLaunchKernelGGL(rt_clear_words_kernel, dim3(1), dim3(128), 0, stream, reinterpret_cast<uint32_t *>(base), RT_STATIC_SAMPLES_AT / 4u);
    RT_HIP(hipGetLastError());
    launch_shade(true, detailed, s->specular, shade_grid, stream, P);
    RT_HIP(hipGetLastError());

    FinishParams F;
    memset(&F, 0, sizeof(F));
    F.S = P.S; F.lists[0] = list; F.lists[1] = list; F.capacity = RT_STATIC_LIST_SLOTS;
    F.rounds = q_shade; F.last_round = 0; F.finish_below = 0xFFFFFFFFu;      /* every parked path is finished here */
    F.totals = s->d_stats + RT_KINDS * RT_STAT_WORDS;
    F.samples = samples; F.stats = s->d_stats + KIND_FINISH * RT_STAT_WORDS;
    F.seed0 = P.seed0; F.seed1 = P.seed1; F.max_segments = P.max_segments; F.half_w = P.half_w; F.half_h = P.half_h;
    uint32_t fgrid = (uint32_t)((paths + 8192u + 255u) / 256u);      /* a workgroup per 256 slots that can hold a record */
    if (fgrid > s->finish_blocks) fgrid = s->finish_blocks;
    const size_t lds = (size_t)RT_BLOCK * s->dev.stack_cap * 4;
    launch_finish(detailed, s->specular, fgrid, lds + RT_FINISH_EXTRA_LDS, stream, F);
    RT_HIP(hipGetLastError());

    ResolveParams R;
    memset(&R, 0, sizeof(R));
    R.samples = samples; R.pixels = device_pixels;
    R.x0 = P.x0; R.y0 = P.y0; R.w = P.w; R.h = P.h;
    R.blocks_x = G.blocks_x; R.block_rows = G.block_rows;
    R.band_blocks = G.band_blocks; R.band_rank = G.band_rank; R.band_count = G.band_count;
    R.compact = (flags & RT_GPU_FLAG_COMPACT_BANDS) ? 1 : 0;
    R.out_pitch = R.compact ? region.size.x : width;
    R.norm = rt_pixel_norm((o && o->spp_total) ? o->spp_total : spp);
    R.launch_spp = spp;
    R.start_from_pixels = accumulate ? 1 : 0;
    R.finish = accumulate ? 0 : 1;
    hipLaunchKernelGGL(rt_resolve_kernel, dim3((uint32_t)((units * 64 + RT_BLOCK - 1) / RT_BLOCK)), dim3(RT_BLOCK), 0, stream, R);
    RT_HIP(hipGetLastError());
    s->launches++;
    s->samples += (uint64_t)spp * own_rows * region.size.x;
    return RT_GPU_OK;
}

static int ensure_frame(rt_gpu_scene *s, size_t floats)
{
    if (s->frame_floats >= floats) return 0;
    if (s->d_frame) { (void)hipFree(s->d_frame); s->d_frame = nullptr; s->frame_floats = 0; }
    if (hipMalloc(reinterpret_cast<void **>(&s->d_frame), floats * sizeof(float)) != hipSuccess)
        return fail(RT_GPU_ERR_MEMORY, "hipMalloc failed for the frame buffer");
    s->frame_floats = floats;
    return 0;
}

int rt_gpu_render_region(rt_gpu_scene *s, rt_image3f *image, rt_rect2u region, uint32_t spp, uint64_t seed,
                         const rt_gpu_options *o)
{
    if (!s || !image || !image->pixels) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_render_region: NULL argument");
    if (o && (o->flags & RT_GPU_FLAG_COMPACT_BANDS)) return fail(RT_GPU_ERR_ARGUMENT, "compact bands are a device-buffer layout");
    if (o && o->struct_size == sizeof(rt_gpu_options) && o->frames > 1) return fail(RT_GPU_ERR_ARGUMENT, "several frames per call need the device-buffer entry point");
    const uint32_t W = image->size.x, H = image->size.y;
    if ((uint64_t)region.point.x + region.size.x > W || (uint64_t)region.point.y + region.size.y > H)
        return fail(RT_GPU_ERR_ARGUMENT, "region exceeds the image");
    if (!region.size.x || !region.size.y) return fail(RT_GPU_ERR_ARGUMENT, "empty region");
    RT_HIP(hipSetDevice(s->device));
    int rc = ensure_frame(s, (size_t)W * H * 3);
    if (rc) return rc;
    /* only the region's rows travel; pixels outside the region keep the host's values */
    const size_t row_bytes = (size_t)region.size.x * 3 * sizeof(float);
    float *d0 = s->d_frame + 3 * ((size_t)W * region.point.y + region.point.x);
    float *h0 = &image->pixels[0].x + 3 * ((size_t)W * region.point.y + region.point.x);
    const size_t pitch = (size_t)W * 3 * sizeof(float);
    if (o && (o->flags & RT_GPU_FLAG_ACCUMULATE))
        RT_HIP(hipMemcpy2D(d0, pitch, h0, pitch, row_bytes, region.size.y, hipMemcpyHostToDevice));
    /* ordered on a stream of the handle's own, and only that stream is waited for: a host that drives other handles or
     * devices from other threads is not held up by their queues (as a device-wide wait would be) */
    rc = rt_gpu_render_region_device(s, s->d_frame, W, H, region, spp, seed, o, s->host_stream);
    if (rc) return rc;
    RT_HIP(hipStreamSynchronize(s->host_stream));
    if (o && o->band_rows) {
        /* copy back only the rows of this rank's bands */
        const uint32_t band = o->band_rows;
        for (uint32_t b = o->band_rank; (uint64_t)b * band < region.size.y; b += o->band_count) {
            uint32_t r0 = b * band, rows = region.size.y - r0 < band ? region.size.y - r0 : band;
            RT_HIP(hipMemcpy2D(h0 + (size_t)r0 * W * 3, pitch, d0 + (size_t)r0 * W * 3, pitch, row_bytes, rows,
                               hipMemcpyDeviceToHost));
        }
    } else {
        RT_HIP(hipMemcpy2D(h0, pitch, d0, pitch, row_bytes, region.size.y, hipMemcpyDeviceToHost));
    }
    return RT_GPU_OK;
}

int rt_gpu_normalize_device(float *device_pixels, size_t pixel_count, uint32_t spp_total, void *hip_stream)
{
    if (!device_pixels || !spp_total) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_normalize_device: bad argument");
    size_t n = pixel_count * 3;
    if (!n) return RT_GPU_OK;
    hipLaunchKernelGGL(rt_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), device_pixels, n, rt_pixel_norm(spp_total));
    RT_HIP(hipGetLastError());
    return RT_GPU_OK;
}

int rt_gpu_quantize_device(const float *device_pixels, size_t pixel_count, uint8_t *device_rgb, void *hip_stream)
{
    if (!device_pixels || !device_rgb) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_quantize_device: NULL argument");
    size_t n = pixel_count * 3;
    if (!n) return RT_GPU_OK;
    if ((n + 255) / 256 > 0x7FFFFFFFull) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_quantize_device: image too large");
    hipLaunchKernelGGL(rt_quantize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(hip_stream), device_pixels, n, device_rgb);
    RT_HIP(hipGetLastError());
    return RT_GPU_OK;
}

int rt_gpu_render_jobs(rt_gpu_scene *s, const rt_job *jobs, uint32_t job_count, const rt_gpu_options *o)
{
    if (!s || !jobs) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_render_jobs: NULL argument");
    if (!job_count) return RT_GPU_OK;
    if (o && o->struct_size != sizeof(rt_gpu_options)) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_options.struct_size mismatch");
    rt_image3f *image = jobs[0].image;
    if (!image || !image->pixels) return fail(RT_GPU_ERR_ARGUMENT, "job without an image");
    const uint32_t W = image->size.x, H = image->size.y;
    std::vector<DevJob> dj(job_count);
    for (uint32_t i = 0; i < job_count; i++) {
        const rt_job &j = jobs[i];
        if (j.image != image) return fail(RT_GPU_ERR_ARGUMENT, "all jobs of one call must share one image");
        if ((uint64_t)j.region.point.x + j.region.size.x > W || (uint64_t)j.region.point.y + j.region.size.y > H)
            return fail(RT_GPU_ERR_ARGUMENT, "job region exceeds the image");
        if (!j.ray_per_pixel) return fail(RT_GPU_ERR_ARGUMENT, "job with zero samples per pixel");
        dj[i].x0 = j.region.point.x; dj[i].y0 = j.region.point.y;
        dj[i].w = j.region.size.x; dj[i].h = j.region.size.y; dj[i].spp = j.ray_per_pixel;
        dj[i].norm = rt_pixel_norm(j.ray_per_pixel);
        memcpy(dj[i].serie, j.serie.bytes, 32);
    }
    RT_HIP(hipSetDevice(s->device));
    int rc = ensure_frame(s, (size_t)W * H * 3);
    if (rc) return rc;
    RT_HIP(hipMemcpy(s->d_frame, &image->pixels[0].x, (size_t)W * H * 12, hipMemcpyHostToDevice));
    DevJob *d_jobs = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d_jobs), sizeof(DevJob) * job_count) != hipSuccess)
        return fail(RT_GPU_ERR_MEMORY, "hipMalloc failed for the job table");
    hipError_t e = hipMemcpy(d_jobs, dj.data(), sizeof(DevJob) * job_count, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(d_jobs); return fail(RT_GPU_ERR_DEVICE, "job upload failed"); }

    JobParams P;
    memset(&P, 0, sizeof(P));
    P.S = s->dev;
    P.S.cam.focal = rt_camera_focal(s->field_of_view, H);
    P.pixels = s->d_frame; P.stats = s->d_stats; P.jobs = d_jobs; P.aes_table = s->d_aes_table;
    P.job_count = job_count; P.width = W; P.height = H;
    P.max_segments = (o && o->max_segments) ? (int)o->max_segments : 64;
    P.half_w = (float)W / 2.0f; P.half_h = (float)H / 2.0f;
    const uint32_t grid = (job_count + RT_BLOCK - 1) / RT_BLOCK;
    const size_t lds = (size_t)RT_BLOCK * s->dev.stack_cap * 4 + 1024;
    hipEvent_t e0 = take_event(s), e1 = take_event(s);
    if (e0 && e1) (void)hipEventRecord(e0, nullptr);
    if (o && (o->flags & RT_GPU_FLAG_DETAILED_STATS))
        hipLaunchKernelGGL(rt_render_jobs_kernel<true>, dim3(grid), dim3(RT_BLOCK), lds, nullptr, P);
    else
        hipLaunchKernelGGL(rt_render_jobs_kernel<false>, dim3(grid), dim3(RT_BLOCK), lds, nullptr, P);
    e = hipGetLastError();
    if (e0 && e1) { (void)hipEventRecord(e1, nullptr); s->timers.push_back({e0, e1, KIND_FRESH}); }
    else { if (e0) s->spare_events.push_back(e0); if (e1) s->spare_events.push_back(e1); }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    (void)hipFree(d_jobs);
    if (e != hipSuccess) return fail(RT_GPU_ERR_DEVICE, std::string("job kernel: ") + hipGetErrorString(e));
    s->launches++;
    RT_HIP(hipMemcpy(&image->pixels[0].x, s->d_frame, (size_t)W * H * 12, hipMemcpyDeviceToHost));
    return RT_GPU_OK;
}

static int find_ray_hit(rt_gpu_scene *s, const rt_ray *rays, uint32_t ray_count, float max_distance,
                        rt_ray_result *out, bool serial)
{
    if (!s || !rays || !out) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_find_ray_hit: NULL argument");
    if (!ray_count) return RT_GPU_OK;
    static_assert(sizeof(rt_ray) == 24 && sizeof(rt_ray_result) == 72, "host record layout");
    RT_HIP(hipSetDevice(s->device));
    float *d_rays = nullptr; uint8_t *d_res = nullptr; uint32_t *d_list = nullptr;
    if (hipMalloc(reinterpret_cast<void **>(&d_rays), (size_t)ray_count * 24) != hipSuccess)
        return fail(RT_GPU_ERR_MEMORY, "hipMalloc failed for the ray table");
    if (hipMalloc(reinterpret_cast<void **>(&d_res), (size_t)ray_count * 72) != hipSuccess) {
        (void)hipFree(d_rays);
        return fail(RT_GPU_ERR_MEMORY, "hipMalloc failed for the result table");
    }
    uint32_t *d_spill = nullptr;
    if (!serial && (hipMalloc(reinterpret_cast<void **>(&d_list), (size_t)ray_count * RT_RECORD_WORDS * 4) != hipSuccess ||
                    hipMalloc(reinterpret_cast<void **>(&d_spill), s->spill_bytes) != hipSuccess)) {
        (void)hipFree(d_rays); (void)hipFree(d_res);
        if (d_list) (void)hipFree(d_list);
        return fail(RT_GPU_ERR_MEMORY, "hipMalloc failed for the ray records");
    }
    const uint32_t grid = (ray_count + RT_BLOCK - 1) / RT_BLOCK;
    const size_t lds = (size_t)RT_BLOCK * s->dev.stack_cap * 4;
    hipError_t e = hipMemcpy(d_rays, rays, (size_t)ray_count * 24, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(d_res, 0, (size_t)ray_count * 72);
    if (e == hipSuccess && serial) {
        HitParams P;
        memset(&P, 0, sizeof(P));
        P.S = s->dev; P.rays = d_rays; P.results = d_res; P.stats = s->d_stats;
        P.count = ray_count; P.max_distance = max_distance;
        hipLaunchKernelGGL(rt_find_hit_kernel, dim3(grid), dim3(RT_BLOCK), lds, nullptr, P);
        e = hipGetLastError();
    } else if (e == hipSuccess) {
        HitPackParams P;
        memset(&P, 0, sizeof(P));
        P.S = s->dev; P.rays = d_rays; P.list = d_list; P.stats = s->d_stats;
        P.count = ray_count; P.max_distance = max_distance;
        hipLaunchKernelGGL(rt_hit_pack_kernel, dim3(grid), dim3(RT_BLOCK), 0, nullptr, P);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemsetAsync(s->d_queue, 0, sizeof(QueueCounters), nullptr);
        if (e == hipSuccess) {
            TreeParams T;
            memset(&T, 0, sizeof(T));
            T.S = s->dev; T.list = d_list; T.capacity = ray_count; T.fixed_slots = ray_count; T.q = s->d_queue; T.stats = s->d_stats;
            T.spill = d_spill; T.stack_levels = s->tree_levels; T.depth_hist = s->d_depth_hist;
            uint32_t tgrid = s->tree_blocks < grid ? s->tree_blocks : grid;      /* the spill area has a slice for every thread of tree_blocks workgroups */
            hipLaunchKernelGGL(rt_tree_kernel<true>, dim3(tgrid), dim3(RT_BLOCK), (size_t)RT_BLOCK * s->tree_levels * 4 + RT_TREE_EXTRA_LDS, nullptr, T);
            e = hipGetLastError();
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(rt_hit_unpack_kernel, dim3(grid), dim3(RT_BLOCK), 0, nullptr, P, d_res);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(out, d_res, (size_t)ray_count * 72, hipMemcpyDeviceToHost);
    (void)hipFree(d_rays); (void)hipFree(d_res);
    if (d_list) (void)hipFree(d_list);
    if (d_spill) (void)hipFree(d_spill);
    if (e != hipSuccess) return fail(RT_GPU_ERR_DEVICE, std::string("closest-hit probe: ") + hipGetErrorString(e));
    return RT_GPU_OK;
}

int rt_gpu_find_ray_hit(rt_gpu_scene *s, const rt_ray *rays, uint32_t ray_count, float max_distance,
                        rt_ray_result *out)
{
    return find_ray_hit(s, rays, ray_count, max_distance, out, false);
}

int rt_gpu_find_ray_hit_serial(rt_gpu_scene *s, const rt_ray *rays, uint32_t ray_count, float max_distance,
                               rt_ray_result *out)
{
    return find_ray_hit(s, rays, ray_count, max_distance, out, true);
}

/* src/main.c:73-82 on the host: min(c,1) (a NaN becomes 1), powf(c, 1/2.2f), *255, truncate */
int rt_gpu_quantize(const float *pixels, size_t pixel_count, uint8_t *rgb)
{
    if (!pixels || !rgb) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_quantize: NULL argument");
    for (size_t i = 0; i < 3 * pixel_count; i++) {
        float c = pixels[i] < 1.0f ? pixels[i] : 1.0f;
        c = powf(c, 1.0f / 2.2f);
        c = c * 255.0f;
        rgb[i] = (uint8_t)c;
    }
    return RT_GPU_OK;
}

static void fill_counters(rt_gpu_stats *out, const uint64_t *w)
{
    out->samples = w[0]; out->rays = w[1]; out->shadow_rays = w[2]; out->root_tests = w[3]; out->interior_pops = w[4];
    out->leaf_pops = w[5]; out->tri_tests = w[6]; out->tri_hits = w[7]; out->tex_lookups = w[8];
    out->rng_draws = w[9]; out->light_tests = w[10]; out->sphere_tests = w[11]; out->plane_tests = w[12];
    out->shade_steps = w[13]; out->pop_steps = w[14]; out->tri_steps = w[15];
    out->shade_lanes = w[16]; out->pop_lanes = w[17]; out->tri_lanes = w[18]; out->tree_entries = w[19];
}

static int read_stats(rt_gpu_scene *s, uint64_t w[RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS + RT_DEPTH_BINS])
{
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(hipDeviceSynchronize());
    settle_timers(s);
    RT_HIP(hipMemcpy(w, s->d_stats, sizeof(uint64_t) * (RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS + RT_DEPTH_BINS), hipMemcpyDeviceToHost));
    return 0;
}

int rt_gpu_get_stats(rt_gpu_scene *s, rt_gpu_stats *out)
{
    if (!s || !out) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_get_stats: NULL argument");
    uint64_t w[RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS + RT_DEPTH_BINS];
    int rc = read_stats(s, w);
    if (rc) return rc;
    uint64_t sum[RT_STAT_WORDS] = {0};
    for (int k = 0; k < RT_KINDS; k++) for (int i = 0; i < RT_STAT_WORDS; i++) sum[i] += w[k * RT_STAT_WORDS + i];
    memset(out, 0, sizeof(*out));
    fill_counters(out, sum);
    if (!out->samples) out->samples = s->samples;
    out->kernel_ms = s->kind_ms[KIND_FRESH] + s->kind_ms[KIND_RESUME] + s->kind_ms[KIND_TREE] + s->kind_ms[KIND_FINISH];
    out->shade_ms = s->kind_ms[KIND_FRESH] + s->kind_ms[KIND_RESUME]; out->tree_ms = s->kind_ms[KIND_TREE];
    out->fresh_ms = s->kind_ms[KIND_FRESH]; out->finish_ms = s->kind_ms[KIND_FINISH];
    out->rounds = w[RT_KINDS * RT_STAT_WORDS]; out->overflowed = w[RT_KINDS * RT_STAT_WORDS + 1];
    out->launches = s->launches; out->passes = s->passes;
    out->passes_with_work = (uint32_t)w[RT_KINDS * RT_STAT_WORDS + 2]; out->lists_full = s->lists_full ? 1u : 0u;
    if (out->overflowed) return fail(RT_GPU_ERR_DEVICE, "parked-path list overflow (" + std::to_string(out->overflowed) + " paths lost): the frames since the last reset are incomplete");
    return RT_GPU_OK;
}

int rt_gpu_get_stats_by_kernel(rt_gpu_scene *s, rt_gpu_stats out[4])
{
    if (!s || !out) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_get_stats_by_kernel: NULL argument");
    uint64_t w[RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS + RT_DEPTH_BINS];
    int rc = read_stats(s, w);
    if (rc) return rc;
    memset(out, 0, sizeof(rt_gpu_stats) * RT_KINDS);
    for (int k = 0; k < RT_KINDS; k++) {
        fill_counters(out + k, w + k * RT_STAT_WORDS);
        out[k].kernel_ms = s->kind_ms[k];
        out[k].launches = s->launches; out[k].passes = s->passes;
        out[k].passes_with_work = (uint32_t)w[RT_KINDS * RT_STAT_WORDS + 2]; out[k].lists_full = s->lists_full ? 1u : 0u;
    }
    out[KIND_FRESH].shade_ms = out[KIND_FRESH].fresh_ms = s->kind_ms[KIND_FRESH];
    out[KIND_RESUME].shade_ms = s->kind_ms[KIND_RESUME];
    out[KIND_TREE].tree_ms = s->kind_ms[KIND_TREE];
    out[KIND_FINISH].finish_ms = s->kind_ms[KIND_FINISH];
    out[KIND_FINISH].rounds = w[RT_KINDS * RT_STAT_WORDS]; out[KIND_FINISH].overflowed = w[RT_KINDS * RT_STAT_WORDS + 1];
    return RT_GPU_OK;
}

int rt_gpu_get_stack_histogram(rt_gpu_scene *s, uint64_t out[32])
{
    if (!s || !out) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_get_stack_histogram: NULL argument");
    static_assert(RT_DEPTH_BINS == 32, "the header promises 32 bins");
    uint64_t w[RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS + RT_DEPTH_BINS];
    int rc = read_stats(s, w);
    if (rc) return rc;
    memcpy(out, w + RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS, sizeof(uint64_t) * RT_DEPTH_BINS);
    return RT_GPU_OK;
}

int rt_gpu_reset_stats(rt_gpu_scene *s)
{
    if (!s) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_reset_stats: NULL argument");
    RT_HIP(hipSetDevice(s->device));
    RT_HIP(hipDeviceSynchronize());
    settle_timers(s);
    RT_HIP(hipMemset(s->d_stats, 0, sizeof(uint64_t) * (RT_KINDS * RT_STAT_WORDS + RT_TOTAL_WORDS + RT_DEPTH_BINS)));
    for (int k = 0; k < RT_KINDS; k++) s->kind_ms[k] = 0.0;
    s->launches = 0; s->passes = 0; s->samples = 0;
    return RT_GPU_OK;
}

int rt_gpu_eval_math(const float *x, const float *y, size_t n, float *sin_out, float *cos_out, float *log_out,
                     float *div_out, float *sqrt_out, float *div_cached_out)
{
    if (!x) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_eval_math: NULL argument");
    if (!n) return RT_GPU_OK;
    float *d[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    const float *hin[2] = {x, y};
    float *hout[6] = {sin_out, cos_out, log_out, div_out, sqrt_out, div_cached_out};
    hipError_t e = hipSuccess;
    for (int i = 0; i < 8 && e == hipSuccess; i++) {
        bool wanted = i < 2 ? hin[i] != nullptr : hout[i - 2] != nullptr;
        if (!wanted) continue;
        e = hipMalloc(reinterpret_cast<void **>(&d[i]), n * sizeof(float));
        if (e == hipSuccess && i < 2) e = hipMemcpy(d[i], hin[i], n * sizeof(float), hipMemcpyHostToDevice);
    }
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt_math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr,
                           d[0], d[1], n, d[2], d[3], d[4], d[5], d[6], d[7]);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    for (int i = 2; i < 8 && e == hipSuccess; i++)
        if (d[i]) e = hipMemcpy(hout[i - 2], d[i], n * sizeof(float), hipMemcpyDeviceToHost);
    for (int i = 0; i < 8; i++) if (d[i]) (void)hipFree(d[i]);
    if (e != hipSuccess) return fail(RT_GPU_ERR_DEVICE, std::string("math probe: ") + hipGetErrorString(e));
    return RT_GPU_OK;
}

int rt_gpu_eval_unit(const float *xyz, size_t n, float *unit_out, float *length_out)
{
    if (!xyz || !unit_out || !length_out) return fail(RT_GPU_ERR_ARGUMENT, "rt_gpu_eval_unit: NULL argument");
    if (!n) return RT_GPU_OK;
    float *d_in = nullptr, *d_unit = nullptr, *d_len = nullptr;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_in), 3 * n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_unit), 3 * n * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_len), n * sizeof(float));
    if (e == hipSuccess) e = hipMemcpy(d_in, xyz, 3 * n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(rt_unit_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, d_in, n, d_unit, d_len);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(unit_out, d_unit, 3 * n * sizeof(float), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(length_out, d_len, n * sizeof(float), hipMemcpyDeviceToHost);
    if (d_in) (void)hipFree(d_in);
    if (d_unit) (void)hipFree(d_unit);
    if (d_len) (void)hipFree(d_len);
    if (e != hipSuccess) return fail(RT_GPU_ERR_DEVICE, std::string("unit probe: ") + hipGetErrorString(e));
    return RT_GPU_OK;
}

} /* extern "C" */
