// Native batch sampler: bit-exact restatement of the reference's sampler.py
// (random_neq sampler.py:9-14, sample_function sampler.py:16-81, WarpSampler sampler.py:83-136)
// and of the integer half of util.py it calls (TimeStamp util.py:24-29, get_timedelta_bin
// util.py:73-120).  Host code only (no HIP): one producer thread per handle fills a bounded ring,
// like the reference's worker process + Queue(maxsize=10).
//
// Random stream = numpy's legacy global generator as the reference drives it:
//   np.random.seed(s)        -> MT19937 init_genrand(s)
//   np.random.randint(l, r)  -> rng = r-1-l; mask = next_pow2(rng)-1 style bit smear;
//                               draw 32-bit words, `& mask`, reject while > rng
// The number of draws is data dependent, so the stream is inherently serial (one producer).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

#include "castrec.h"

namespace {

struct MT19937 {
    uint32_t mt[624];
    int idx;
    void seed(uint32_t s) {
        mt[0] = s;
        for (int i = 1; i < 624; ++i) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (uint32_t)i;
        idx = 624;
    }
    void gen() {
        for (int k = 0; k < 624; ++k) {
            uint32_t y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
            mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        idx = 0;
    }
    uint32_t next() {
        if (idx >= 624) gen();
        uint32_t y = mt[idx++];
        y ^= (y >> 11);
        y ^= (y << 7) & 0x9d2c5680u;
        y ^= (y << 15) & 0xefc60000u;
        y ^= (y >> 18);
        return y;
    }
    // numpy legacy RandomState.randint(low, high) for ranges < 2^32
    int64_t randint(int64_t low, int64_t high) {
        uint64_t rng = (uint64_t)(high - 1 - low);
        if (rng == 0) return low;
        uint64_t mask = rng;
        mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16; mask |= mask >> 32;
        if (rng == 0xFFFFFFFFull) return low + (int64_t)next();
        uint32_t v;
        do { v = next() & (uint32_t)mask; } while (v > rng);
        return low + (int64_t)v;
    }
};

inline int64_t floordiv(int64_t a, int64_t b) {   // Python // for b > 0
    int64_t q = a / b;
    if ((a % b != 0) && (a < 0)) --q;
    return q;
}

struct Batch {
    std::vector<int32_t> user, seq, pos, neg, timeseq, ratings, hours, days;
    void init(int B, int T) {
        user.assign(B, 0);
        for (auto* v : {&seq, &pos, &neg, &timeseq, &ratings, &hours, &days}) v->assign((size_t)B * T, 0);
    }
};

}  // namespace

struct cr_sampler {
    std::vector<int64_t> offsets, ts;
    std::vector<int32_t> items;
    std::vector<float> ratings;
    int usernum, itemnum, B, T, bin_in_hours, max_bins, log_scale;
    double min_td, max_td;
    MT19937 rng;
    std::vector<uint32_t> stamp;
    uint32_t epoch = 0;
    // ring
    std::vector<Batch> ring;
    size_t head = 0, tail = 0, count = 0;
    bool stop = false;
    std::mutex mu;
    std::condition_variable cv_full, cv_empty;
    std::thread worker;

    int timebin(int64_t delta) const {
        if (log_scale) {   // sampler.py:66 passes bin_in_hours=48, max_bins=200 literally; util.py:95-109
            const double lo = min_td + 1.0, hi = max_td + 1.0, t = (double)delta + 1.0;
            const double bin_size = (log(hi) - log(lo)) / 200.0;
            double b = floor(log(t) / bin_size);
            if (b > 200.0) b = 200.0;
            return (int)b;
        }
        // util.py:114: floor(ts // 3600 / bin_in_hours) == ts // (3600*bin_in_hours) for integers
        int64_t b = floordiv(floordiv(delta, 3600), (int64_t)bin_in_hours);
        if (b > max_bins) b = max_bins;   // util.py:117-118
        return (int)b;
    }

    void sample(Batch& out, int row) {
        // sampler.py:19-21
        int64_t user = rng.randint(1, (int64_t)usernum + 1);
        while (offsets[user + 1] - offsets[user] <= 1) user = rng.randint(1, (int64_t)usernum + 1);
        const int64_t a = offsets[user], b = offsets[user + 1];
        int32_t* seq = &out.seq[(size_t)row * T];
        int32_t* pos = &out.pos[(size_t)row * T];
        int32_t* neg = &out.neg[(size_t)row * T];
        int32_t* tsq = &out.timeseq[(size_t)row * T];
        int32_t* rat = &out.ratings[(size_t)row * T];
        int32_t* hrs = &out.hours[(size_t)row * T];
        int32_t* dys = &out.days[(size_t)row * T];
        memset(seq, 0, sizeof(int32_t) * T); memset(pos, 0, sizeof(int32_t) * T); memset(neg, 0, sizeof(int32_t) * T);
        memset(tsq, 0, sizeof(int32_t) * T); memset(rat, 0, sizeof(int32_t) * T);
        memset(hrs, 0, sizeof(int32_t) * T); memset(dys, 0, sizeof(int32_t) * T);
        out.user[row] = (int32_t)user;
        // ts = set(items of user)  (sampler.py:42) as an epoch-stamped membership table
        if (++epoch == 0) { std::fill(stamp.begin(), stamp.end(), 0u); epoch = 1; }
        for (int64_t e = a; e < b; ++e) stamp[items[e]] = epoch;
        int32_t nxt = items[b - 1];
        int idx = T - 1;
        int first = T;                       // first filled slot
        for (int64_t e = b - 2; e >= a; --e) {   // reversed(user_train[user][:-1])  (sampler.py:44)
            seq[idx] = items[e];
            rat[idx] = (int32_t)ratings[e];
            hrs[idx] = (int32_t)(floordiv(ts[e], 3600) % 24 + 1);          // util.py:28 (UTC hour + 1)
            dys[idx] = (int32_t)((floordiv(ts[e], 86400) + 3) % 7 + 1);    // util.py:27 (ISO weekday)
            pos[idx] = nxt;
            if (nxt != 0) {                  // sampler.py:52-55 random_neq(1, itemnum+1, ts)
                int64_t t = rng.randint(1, (int64_t)itemnum + 1);
                while (stamp[t] == epoch) t = rng.randint(1, (int64_t)itemnum + 1);
                neg[idx] = (int32_t)t;
            }
            nxt = items[e];
            first = idx;
            --idx;
            if (idx == -1) break;
        }
        // sampler.py:61-72: bins relative to the most recent item of the (truncated) window
        const int64_t most_recent = ts[b - 2];
        int64_t e = b - 2;
        for (int i = T - 1; i >= first; --i, --e) tsq[i] = timebin(most_recent - ts[e]);
    }

    void run() {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_full.wait(lk, [&] { return stop || count < ring.size(); });
                if (stop) return;
            }
            Batch& bt = ring[tail];
            for (int r = 0; r < B; ++r) sample(bt, r);
            {
                std::lock_guard<std::mutex> lk(mu);
                tail = (tail + 1) % ring.size();
                ++count;
            }
            cv_empty.notify_one();
        }
    }
};

extern "C" cr_sampler* cr_sampler_create(const int64_t* offsets, const int32_t* items, const float* ratings,
                                         const int64_t* ts, int usernum, int itemnum, int batch_size, int maxlen,
                                         int bin_in_hours, int max_bins, int log_scale, double min_timedelta,
                                         double max_timedelta, uint32_t seed, int queue_depth) {
    if (!offsets || !items || !ts || usernum <= 0 || itemnum <= 0 || batch_size <= 0 || maxlen <= 0 || bin_in_hours <= 0)
        return nullptr;
    const int64_t nnz = offsets[usernum + 1];
    bool any = false;
    for (int u = 1; u <= usernum; ++u) any |= (offsets[u + 1] - offsets[u] > 1);
    if (!any) return nullptr;               // sampler.py:20 would spin forever
    for (int64_t e = 0; e < nnz; ++e)
        if (items[e] < 0 || items[e] > itemnum) return nullptr;
    cr_sampler* s = new cr_sampler();
    s->offsets.assign(offsets, offsets + usernum + 2);
    s->items.assign(items, items + nnz);
    s->ts.assign(ts, ts + nnz);
    if (ratings) s->ratings.assign(ratings, ratings + nnz); else s->ratings.assign(nnz, 0.0f);
    s->usernum = usernum; s->itemnum = itemnum; s->B = batch_size; s->T = maxlen;
    s->bin_in_hours = bin_in_hours; s->max_bins = max_bins; s->log_scale = log_scale;
    s->min_td = min_timedelta; s->max_td = max_timedelta;
    s->rng.seed(seed);                      // np.random.seed(SEED)  (sampler.py:76)
    s->stamp.assign((size_t)itemnum + 1, 0u);
    if (queue_depth < 1) queue_depth = 10;  // Queue(maxsize=n_workers*10)  (sampler.py:103)
    s->ring.resize(queue_depth);
    for (auto& b : s->ring) b.init(batch_size, maxlen);
    s->worker = std::thread([s] { s->run(); });
    return s;
}

extern "C" int cr_sampler_next(cr_sampler* s, int32_t* user, int32_t* seq, int32_t* pos, int32_t* neg,
                               int32_t* timeseq, int32_t* ratings, int32_t* hours, int32_t* days) {
    if (!s) return CR_ERR_INVALID;
    {
        std::unique_lock<std::mutex> lk(s->mu);
        s->cv_empty.wait(lk, [&] { return s->count > 0; });
    }
    const Batch& b = s->ring[s->head];
    const size_t n = (size_t)s->B * s->T * sizeof(int32_t);
    if (user) memcpy(user, b.user.data(), sizeof(int32_t) * s->B);
    if (seq) memcpy(seq, b.seq.data(), n);
    if (pos) memcpy(pos, b.pos.data(), n);
    if (neg) memcpy(neg, b.neg.data(), n);
    if (timeseq) memcpy(timeseq, b.timeseq.data(), n);
    if (ratings) memcpy(ratings, b.ratings.data(), n);
    if (hours) memcpy(hours, b.hours.data(), n);
    if (days) memcpy(days, b.days.data(), n);
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->head = (s->head + 1) % s->ring.size();
        --s->count;
    }
    s->cv_full.notify_one();
    return CR_OK;
}

extern "C" void cr_sampler_destroy(cr_sampler* s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
    }
    s->cv_full.notify_all();
    if (s->worker.joinable()) s->worker.join();
    delete s;
}
