// Host-side state of the CoVA filter elements behind the C-ABI (include/covahip.h):
// Bbox / Frame bincode wire format, metapreprocess stacking ring, SORT tracker
// (Kalman + Hungarian) and the cova GoP frame filter.  Plain C++17, no GPU.
//
// Reference (paths under /root/reference):
//   cova-rs/bbox/src/bbox.rs:4-91, cova-rs/bbox/src/lib.rs:8-22
//   cova-rs/gst-plugins/src/metapreprocess/imp.rs:204-332
//   cova-rs/sort/src/lib.rs:14-214, state.rs:9-28, tracker/mod.rs:15-152,
//   tracker/motion_model.rs:38-55, tracker/linear_observation_model.rs:33-47
//   cova-rs/gst-plugins/src/cova/imp.rs:90-432, cova/tracker.rs:16-125
// Third-party arithmetic restated from the published algorithms (sources absent):
//   adskalman 0.13.0 (Kalman predict / Joseph-form update), linear_assignment 0.0.2
//   (min-cost perfect matching), bincode 1.3.3 default config (LE, fixed-width ints,
//   u64 length prefix, 1-byte Option tag).
#include <algorithm>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <limits>
#include <list>
#include <map>
#include <new>
#include <memory>
#include <vector>

#include "covahip.h"

// ============================================================== bincode
namespace {

struct Writer {
    uint8_t *out;
    size_t cap;
    size_t n = 0;
    void put(const void *p, size_t len) {
        if (out && n + len <= cap) std::memcpy(out + n, p, len);
        n += len;
    }
    template <typename T>
    void val(T v) { put(&v, sizeof(T)); }
};

void write_bbox(Writer &w, const covahip_bbox &b) {
    w.val<float>(b.left);
    w.val<float>(b.top);
    w.val<float>(b.width);
    w.val<float>(b.height);
    w.val<float>(b.area);
    w.val<uint8_t>(b.has_track_id ? 1 : 0);
    if (b.has_track_id) w.val<uint64_t>(b.track_id);
    w.val<uint8_t>(b.has_timestamp ? 1 : 0);
    if (b.has_timestamp) w.val<uint64_t>(b.timestamp);
    w.val<uint8_t>(b.has_class_id ? 1 : 0);
    if (b.has_class_id) w.val<uint32_t>(b.class_id);
    w.val<uint8_t>(b.has_confidence ? 1 : 0);
    if (b.has_confidence) w.val<float>(b.confidence);
}

struct Reader {
    const uint8_t *p;
    size_t len;
    size_t n = 0;
    bool ok = true;
    template <typename T>
    T val() {
        T v{};
        if (n + sizeof(T) > len) { ok = false; return v; }
        std::memcpy(&v, p + n, sizeof(T));
        n += sizeof(T);
        return v;
    }
};

bool read_bbox(Reader &r, covahip_bbox &b) {
    std::memset(&b, 0, sizeof(b));
    b.left = r.val<float>();
    b.top = r.val<float>();
    b.width = r.val<float>();
    b.height = r.val<float>();
    b.area = r.val<float>();
    uint8_t t = r.val<uint8_t>();
    if (t > 1) return false;
    b.has_track_id = t;
    if (t) b.track_id = r.val<uint64_t>();
    t = r.val<uint8_t>();
    if (t > 1) return false;
    b.has_timestamp = t;
    if (t) b.timestamp = r.val<uint64_t>();
    t = r.val<uint8_t>();
    if (t > 1) return false;
    b.has_class_id = t;
    if (t) b.class_id = r.val<uint32_t>();
    t = r.val<uint8_t>();
    if (t > 1) return false;
    b.has_confidence = t;
    if (t) b.confidence = r.val<float>();
    return r.ok;
}

covahip_bbox bbox_new(float l, float t, float w, float h) {  // Bbox::new, bbox.rs:17-29
    covahip_bbox b;
    std::memset(&b, 0, sizeof(b));
    b.left = l;
    b.top = t;
    b.width = w;
    b.height = h;
    b.area = w * h;
    return b;
}

float bbox_iou(const covahip_bbox &s, const covahip_bbox &t) {  // bbox.rs:39-56
    const float s_x2 = s.left + s.width, s_y2 = s.top + s.height;
    const float t_x2 = t.left + t.width, t_y2 = t.top + t.height;
    const float x_left = std::fmax(s.left, t.left), y_top = std::fmax(s.top, t.top);
    const float x_right = std::fmin(s_x2, t_x2), y_bottom = std::fmin(s_y2, t_y2);
    if (x_right <= x_left || y_bottom <= y_top) return 0.f;
    const float inter = (x_right - x_left) * (y_bottom - y_top);
    const float uni = s.area + t.area - inter;
    return inter / uni;
}

}  // namespace

extern "C" {

void covahip_boxes_to_bbox(const covahip_box *in, int n, covahip_bbox *out) {
    for (int i = 0; i < n; i++)
        out[i] = bbox_new((float)in[i].left, (float)in[i].top, (float)in[i].width, (float)in[i].height);
}

size_t covahip_bbox_serialize_vec(const covahip_bbox *boxes, size_t n, uint8_t *out, size_t cap, int *status) {
    Writer w{out, cap};
    w.val<uint64_t>((uint64_t)n);
    for (size_t i = 0; i < n; i++) write_bbox(w, boxes[i]);
    if (status) *status = (out && w.n <= cap) ? COVAHIP_OK : COVAHIP_ERR_OVERFLOW;
    return w.n;
}

int covahip_bbox_deserialize_vec(const uint8_t *data, size_t len, covahip_bbox *out, size_t cap, size_t *n) {
    if (!data || !n) return COVAHIP_ERR_INVALID_ARG;
    Reader r{data, len};
    const uint64_t cnt = r.val<uint64_t>();
    if (!r.ok) return COVAHIP_ERR_BAD_DATA;
    if (cnt > len) return COVAHIP_ERR_BAD_DATA;  // each box is >= 24 bytes
    *n = (size_t)cnt;
    for (uint64_t i = 0; i < cnt; i++) {
        covahip_bbox b;
        if (!read_bbox(r, b)) return COVAHIP_ERR_BAD_DATA;
        if (out && i < cap) out[i] = b;
    }
    if (r.n != len) return COVAHIP_ERR_BAD_DATA;  // trailing bytes
    return (cnt > cap && out) ? COVAHIP_ERR_OVERFLOW : COVAHIP_OK;
}

size_t covahip_frame_serialize(uint64_t range_start, uint64_t oldest, const covahip_bbox *boxes, size_t n,
                               uint8_t *out, size_t cap, int *status) {
    Writer w{out, cap};
    w.val<uint64_t>(range_start);
    w.val<uint64_t>(oldest);
    w.val<uint64_t>((uint64_t)n);
    for (size_t i = 0; i < n; i++) write_bbox(w, boxes[i]);
    if (status) *status = (out && w.n <= cap) ? COVAHIP_OK : COVAHIP_ERR_OVERFLOW;
    return w.n;
}

float covahip_bbox_iou(const covahip_bbox *a, const covahip_bbox *b) { return bbox_iou(*a, *b); }

}  // extern "C"

// ============================================================== metapreprocess ring
struct covahip_stack {
    size_t size_per_buf;
    unsigned timestep, gamma;
    size_t gamma_idx = 0;
    std::deque<std::vector<uint8_t>> prev;  // front = newest (imp.rs:304,321)
};

extern "C" {

int covahip_stack_new(size_t size_per_buf, unsigned timestep, unsigned gamma, covahip_stack **out) {
    if (!out || timestep < 1 || gamma < 1 || size_per_buf == 0) return COVAHIP_ERR_INVALID_ARG;
    covahip_stack *s = new (std::nothrow) covahip_stack();
    if (!s) return COVAHIP_ERR_INVALID_ARG;
    s->size_per_buf = size_per_buf;
    s->timestep = timestep;
    s->gamma = gamma;
    *out = s;
    return COVAHIP_OK;
}

void covahip_stack_free(covahip_stack *s) { delete s; }

int covahip_stack_push(covahip_stack *s, const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                       int *emitted) {
    if (!s || !in || !emitted) return COVAHIP_ERR_INVALID_ARG;
    if (in_len < s->size_per_buf) return COVAHIP_ERR_INVALID_ARG;
    *emitted = 0;
    const size_t spb = s->size_per_buf;
    if (s->prev.size() < (size_t)s->timestep - 1) {  // imp.rs:302-305: warm-up, FLOW_DROPPED
        s->prev.emplace_front(in, in + spb);
        return COVAHIP_OK;
    }
    if (s->gamma_idx == 0) {  // imp.rs:306-324
        if (!out || out_cap < spb * s->timestep) return COVAHIP_ERR_OVERFLOW;
        std::memcpy(out, in, spb);
        size_t idx = spb;
        for (const auto &p : s->prev) {
            std::memcpy(out + idx, p.data(), spb);
            idx += spb;
        }
        *emitted = 1;
        s->gamma_idx = s->gamma - 1;
    } else {  // imp.rs:325-330
        s->gamma_idx -= 1;
    }
    s->prev.emplace_front(in, in + spb);
    s->prev.pop_back();
    return COVAHIP_OK;
}

void covahip_stack_out_dims(int width, int height, unsigned timestep, int *out_w, int *out_h) {
    if (out_w) *out_w = width / 16;
    if (out_h) *out_h = height / 16 * (int)timestep;
}

}  // extern "C"

// ============================================================== SORT
namespace {

typedef float P;  // PrecisionType = f32 (sort/src/lib.rs:3)

struct Mat7 {
    P m[7][7];
};

// x' = F x with F = I + (x0 += x4, x1 += x5, x2 += x6)  (motion_model.rs:38-45,
// nalgebra from_vec is column-major, so the literal is F transposed on the page); P' = F P F^T + Q.
// The products are written out over F's non-zero entries in the order the dense row x column sums visit them (k ascending:
// the diagonal 1, then the 1 at column k + 4 for the first three rows): a product with a zero entry adds +-0 and one with a 1
// is exact, so the results are those of the dense 7x7x7 products bit for bit -- at 49 + 49 additions instead of 686
// multiply-adds (SORT at the experiment's parameters updates dozens of young trackers per frame and stream).
void kalman_predict(const P x[7], const Mat7 &Pm, P xo[7], Mat7 &Po) {
    static const P Q[7] = {1.f, 1.f, 1.f, 1.f, 0.01f, 0.01f, 0.0001f};  // motion_model.rs:48-55
    for (int i = 0; i < 7; i++) xo[i] = i < 3 ? (0.f + x[i]) + x[i + 4] : 0.f + x[i];
    P FP[7][7];
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) FP[i][j] = i < 3 ? (0.f + Pm.m[i][j]) + Pm.m[i + 4][j] : 0.f + Pm.m[i][j];
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) {
            const P a = j < 3 ? (0.f + FP[i][j]) + FP[i][j + 4] : 0.f + FP[i][j];   // * F^T
            Po.m[i][j] = a + (i == j ? Q[i] : 0.f);
        }
}

// adskalman ObservationModel::update, CovarianceUpdateMethod::JosephForm, with
// H = [I4 0] (linear_observation_model.rs:33-40), R = diag(1,1,10,10) (:43-47).
// Returns false when S is not positive definite (adskalman returns Err).
// A = I - K H differs from the identity in its first four columns only: the sums over k stop at 4 and pick up the one
// remaining unit entry at its place in the k order (same values as the dense sums, see kalman_predict).
bool kalman_update(const P xp[7], const Mat7 &Pp, const P z[4], P xo[7], Mat7 &Po) {
    static const P R[4] = {1.f, 1.f, 10.f, 10.f};
    // S = H P H^T + R = P[0:4,0:4] + R
    P S[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) S[i][j] = Pp.m[i][j] + (i == j ? R[i] : 0.f);
    // Cholesky S = L L^T
    P L[4][4] = {};
    for (int j = 0; j < 4; j++) {
        P d = S[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k] * L[j][k];
        if (!(d > 0.f)) return false;
        L[j][j] = std::sqrt(d);
        for (int i = j + 1; i < 4; i++) {
            P a = S[i][j];
            for (int k = 0; k < j; k++) a -= L[i][k] * L[j][k];
            L[i][j] = a / L[j][j];
        }
    }
    // S^-1 via L^-1
    P Li[4][4] = {};
    for (int i = 0; i < 4; i++) {
        Li[i][i] = 1.f / L[i][i];
        for (int j = 0; j < i; j++) {
            P a = 0.f;
            for (int k = j; k < i; k++) a -= L[i][k] * Li[k][j];
            Li[i][j] = a / L[i][i];
        }
    }
    P Si[4][4];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++) {
            P a = 0.f;
            for (int k = 0; k < 4; k++) a += Li[k][i] * Li[k][j];
            Si[i][j] = a;
        }
    // K = P H^T S^-1  (7x4): P H^T = first 4 columns of P
    P K[7][4];
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 4; j++) {
            P a = 0.f;
            for (int k = 0; k < 4; k++) a += Pp.m[i][k] * Si[k][j];
            K[i][j] = a;
        }
    P innov[4];
    for (int i = 0; i < 4; i++) innov[i] = z[i] - xp[i];
    for (int i = 0; i < 7; i++) {
        P a = 0.f;
        for (int k = 0; k < 4; k++) a += K[i][k] * innov[k];
        xo[i] = xp[i] + a;
    }
    // (I - K H) P (I - K H)^T + K R K^T
    P A4[7][4];   // the first four columns of A = I - K H; columns 4..6 are the identity's
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 4; j++) A4[i][j] = (i == j ? 1.f : 0.f) - K[i][j];
    P AP[7][7];
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) {
            P a = 0.f;
            for (int k = 0; k < 4; k++) a += A4[i][k] * Pp.m[k][j];
            if (i >= 4) a += Pp.m[i][j];
            AP[i][j] = a;
        }
    P KR[7][4];
    for (int i = 0; i < 7; i++)
        for (int k = 0; k < 4; k++) KR[i][k] = K[i][k] * R[k];
    for (int i = 0; i < 7; i++)
        for (int j = 0; j < 7; j++) {
            P a = 0.f;
            for (int k = 0; k < 4; k++) a += AP[i][k] * A4[j][k];
            if (j >= 4) a += AP[i][j];
            P b = 0.f;
            for (int k = 0; k < 4; k++) b += KR[i][k] * K[j][k];
            Po.m[i][j] = a + b;
        }
    return true;
}

void into_z(const covahip_bbox &b, P z[4]) {  // state.rs:10-16
    z[0] = b.left + b.width / 2.f;
    z[1] = b.top + b.height / 2.f;
    z[2] = b.area;
    z[3] = b.width / b.height;
}

covahip_bbox from_x(const P x[7]) {  // state.rs:18-28 -- `top` uses width (reference quirk)
    const P r = x[3], s = x[2], y = x[1], xx = x[0];
    const P width = std::sqrt(s * r);
    const P height = s / width;
    return bbox_new(xx - width / 2.f, y - width / 2.f, width, height);
}

struct TrackerBody {  // KalmanBoxTracker, tracker/mod.rs:15-69
    uint64_t id = 0, start = 0, last_match = 0;
    std::vector<uint64_t> seen_ts;
    bool active = false;
    // The predictions of consecutive UNMATCHED frames are the same box with another timestamp (the prior is kept, below), and at
    // the experiment's parameters ~125 of a stream's ~130 trackers are in that state: predict() then only notes the timestamp,
    // and the boxes are written out when somebody looks (hist(): a match, the export of a finished track, the introspection
    // calls).  Same history, entry for entry.
    mutable std::vector<covahip_bbox> history;
    mutable std::vector<uint64_t> pending_ts;
    uint64_t hits = 0, time_since_update = 0, hit_streaks = 0, age = 0;
    // The filter state lives in its own block: Sort::update's retain moves every tracker behind the first one that died -- at
    // the experiment's parameters ~130 trackers per stream and frame -- and a tracker with its two 7x7 matrices inline is 600
    // bytes to move, with the state out of line a few pointers and counters.
    struct KState {
        P x[7];
        Mat7 Pm;  // previous_estimate
        P xp[7];
        Mat7 Pp;  // prior
        covahip_bbox prior_box;
    };
    std::unique_ptr<KState> k{new KState()};
    bool has_prior = false;
    // predict() derives the prior from the previous ESTIMATE (x, Pm), which only a matched update() changes: a tracker that
    // goes unmatched recomputes the same prior -- and the same predicted box -- frame after frame.  At the experiment's
    // parameters (maxage 60, minhits 30) a stream carries dozens of such trackers, and their 7x7 covariance products were
    // 44 % of the element's host time: the prior is kept while the estimate has not changed (same values, bit for bit).
    bool prior_valid = false;

    TrackerBody(uint64_t id_, const covahip_bbox &b, uint64_t start_) : id(id_), start(start_), last_match(start_) {
        P z[4];
        into_z(b, z);
        for (int i = 0; i < 7; i++) k->x[i] = i < 4 ? z[i] : 0.f;
        for (int i = 0; i < 7; i++)
            for (int j = 0; j < 7; j++) k->Pm.m[i][j] = (i == j) ? (i < 4 ? 10.f : 10000.f) : 0.f;
    }

    void flush() const {
        for (uint64_t ts : pending_ts) {
            covahip_bbox b = k->prior_box;
            b.timestamp = ts;
            history.push_back(b);
        }
        pending_ts.clear();
    }
    const std::vector<covahip_bbox> &hist() const { flush(); return history; }
    size_t hist_len() const { return history.size() + pending_ts.size(); }

    bool predict(uint64_t ts) {  // tracker/mod.rs:104-121; the predicted box is k->prior_box with this timestamp.  true: the prior moved
        const bool moved = !prior_valid;
        if (!prior_valid) {      // (nothing is pending here: the estimate only moves in update(), which writes the pending boxes out)
            if (k->x[6] + k->x[2] <= 0.f) k->x[6] = 0.f;
            kalman_predict(k->x, k->Pm, k->xp, k->Pp);
            has_prior = true;
            k->prior_box = from_x(k->xp);
            k->prior_box.has_track_id = 1;
            k->prior_box.track_id = id;
            k->prior_box.has_timestamp = 1;
            prior_valid = true;
        }
        age += 1;
        time_since_update += 1;
        pending_ts.push_back(ts);
        return moved;
    }

    bool update(const covahip_bbox *det) {  // tracker/mod.rs:71-102
        if (det) {
            hits += 1;
            hit_streaks += 1;
            if (hit_streaks >= 5) {  // reference "FIXME: arbitrary number"
                time_since_update = 0;
                last_match = det->timestamp;
            }
            P z[4];
            into_z(*det, z);
            if (!has_prior) return false;
            P xn[7];
            Mat7 Pn;
            if (!kalman_update(k->xp, k->Pp, z, xn, Pn)) return false;
            flush();               // this frame's prediction becomes a box of its own (class and confidence below)
            std::memcpy(k->x, xn, sizeof(k->x));
            k->Pm = Pn;
            prior_valid = false;   // the estimate has moved
            covahip_bbox &last = history.back();
            last.has_class_id = det->has_class_id;
            last.class_id = det->class_id;
            last.has_confidence = det->has_confidence;
            last.confidence = det->confidence;
        } else {
            hit_streaks = 0;
        }
        return true;
    }

    bool should_live(uint64_t max_age) const { return time_since_update <= max_age; }
    void check_activate(uint64_t min_hits) {
        if (!active && hit_streaks >= min_hits) active = true;
    }
    bool is_seen() const {  // tracker/mod.rs:138-142
        for (uint64_t ts : seen_ts)
            if (start <= ts && last_match >= ts) return true;
        return false;
    }
    void trim_dead_history() {  // tracker/mod.rs:144-151: the last time_since_update entries go (mostly pending ones)
        if (time_since_update > hist_len()) return;   // (drop_idx wraps in the reference's u64 arithmetic: nothing is dropped)
        size_t drop = (size_t)time_since_update;
        const size_t from_pending = std::min(drop, pending_ts.size());
        pending_ts.resize(pending_ts.size() - from_pending);
        drop -= from_pending;
        if (drop) { flush(); history.resize(history.size() - drop); }
    }
};

// What Sort keeps in its vector: a handle.  The retain of Sort::update closes the gaps the dead trackers leave -- at the
// experiment's parameters two trackers die per frame near the front of ~130 -- and moving a tracker with its three vectors cost
// 25 ns apiece; a handle is a pointer.
struct Tracker {
    std::unique_ptr<TrackerBody> b;
    Tracker(uint64_t id, const covahip_bbox &box, uint64_t start) : b(new TrackerBody(id, box, start)) {}
    TrackerBody *operator->() const { return b.get(); }
};

// Min-cost assignment of every row of an n x m matrix (n <= m) to a distinct column (shortest augmenting paths with
// potentials, O(n^2 m)); a[i][j] row-major.  Returns col_of_row.
std::vector<int> assign_rows(const std::vector<double> &a, int n, int m) {
    const double INF = std::numeric_limits<double>::infinity();
    // scratch kept per thread: the tracker calls this once per frame and stream, and six allocations were a fifth of its time
    static thread_local std::vector<double> u, v, minv;
    static thread_local std::vector<int> p, way;
    static thread_local std::vector<char> used;
    u.assign(n + 1, 0.0); v.assign(m + 1, 0.0); minv.resize(m + 1);
    p.assign(m + 1, 0); way.assign(m + 1, 0);
    used.resize(m + 1);
    for (int i = 1; i <= n; i++) {
        p[0] = i;
        int j0 = 0;
        std::fill(minv.begin(), minv.end(), INF);
        std::fill(used.begin(), used.end(), 0);
        do {
            used[j0] = 1;
            const int i0 = p[j0];
            double delta = INF;
            int j1 = 0;
            for (int j = 1; j <= m; j++)
                if (!used[j]) {
                    const double cur = a[(size_t)(i0 - 1) * m + (j - 1)] - u[i0] - v[j];
                    if (cur < minv[j]) { minv[j] = cur; way[j] = j0; }
                    if (minv[j] < delta) { delta = minv[j]; j1 = j; }
                }
            for (int j = 0; j <= m; j++)
                if (used[j]) { u[p[j]] += delta; v[j] -= delta; }
                else minv[j] -= delta;
            j0 = j1;
        } while (p[j0] != 0);
        do {
            const int j1 = way[j0];
            p[j0] = p[j1];
            j0 = j1;
        } while (j0);
    }
    std::vector<int> col_of_row(n, -1);
    for (int j = 1; j <= m; j++)
        if (p[j] > 0) col_of_row[p[j] - 1] = j - 1;
    return col_of_row;
}

// linear_assignment() of sort/src/lib.rs:25-56: zero-pad to square, solve, drop padded edges and edges whose ORIGINAL
// cost equals 2.0.  The padded square problem is solved in its rectangular form: with zero-cost pad columns (rows) a
// perfect matching of the square matrix is an assignment of every element of the SHORTER side to a distinct element of the
// longer one plus pad edges of cost 0 -- same feasible set, same objective, so an optimum of one is an optimum of the other.
// With sixty young trackers and four detections (the experiment's minhits 30 / maxage 60 keep that many alive) that is
// 4 x 4 x 60 steps instead of 60^3: 780 -> 3 us per frame at 80 x 4.  Which optimum comes out on exact ties differs between
// solvers either way ("Assignment ties").
std::vector<std::pair<size_t, size_t>> linear_assignment(const std::vector<P> &cost_colmajor, size_t n_rows,
                                                         size_t n_cols) {
    std::vector<std::pair<size_t, size_t>> out;
    if (n_rows == 0 || n_cols == 0) return out;
    const bool by_rows = n_rows <= n_cols;           // the side that is assigned completely
    const size_t n = by_rows ? n_rows : n_cols, m = by_rows ? n_cols : n_rows;
    static thread_local std::vector<double> a;
    a.resize(n * m);
    for (size_t i = 0; i < n_rows; i++)
        for (size_t j = 0; j < n_cols; j++) {
            const P c = cost_colmajor[j * n_rows + i];
            // OrderedFloat sorts NaN above everything; use a large finite stand-in.
            const double cv = std::isnan(c) ? 1e30 : (double)c;
            if (by_rows) a[i * m + j] = cv;
            else a[j * m + i] = cv;
        }
    const std::vector<int> other = assign_rows(a, (int)n, (int)m);
    for (size_t k = 0; k < n; k++) {
        if (other[k] < 0) continue;
        const size_t i = by_rows ? k : (size_t)other[k], j = by_rows ? (size_t)other[k] : k;
        if (cost_colmajor[j * n_rows + i] == 2.0f) continue;
        out.emplace_back(i, j);
    }
    std::sort(out.begin(), out.end());
    return out;
}

struct Sort {  // sort/src/lib.rs:14-23
    uint64_t max_age, min_hits;
    P iou_threshold;
    std::vector<Tracker> trackers;
    uint64_t frame_count = 0, id_counter = 0;
    std::vector<long> match_of_trk;        // scratch of update()
    std::vector<uint8_t> det_matched;

    // scratch of match_dets (kept between frames: a frame allocates nothing here)
    mutable std::vector<P> cost_, red_, px1_, py1_, px2_, py2_, pa_, pw_;
    mutable std::vector<uint8_t> ov_;
    mutable std::vector<size_t> keep_;
    bool soa_stale = false;   // trackers were reordered or stepped outside update(): the next update() rewrites every array entry

    // the predictions are the trackers' prior boxes (Tracker::predict has run for every tracker of this frame)
    std::vector<std::pair<size_t, size_t>> match_dets(const std::vector<covahip_bbox> &dets) const {
        std::vector<std::pair<size_t, size_t>> res;
        const size_t np = trackers.size(), nd = dets.size();
        if (np == 0 || nd == 0) return res;
        // the predictions as arrays: the IoU of every (detection, prediction) pair -- n_dets x ~130 per frame at the experiment's
        // parameters -- is then a loop the compiler vectorises; same expressions in the same order as bbox_iou (bbox.rs:39-56)
        cost_.resize(np * nd);  // column-major: rows = predictions, cols = detections
        ov_.assign(np, 0);
        std::vector<P> &cost = cost_;
        for (size_t j = 0; j < nd; j++) {
            const covahip_bbox &d = dets[j];
            const P dx1 = d.left, dy1 = d.top, dx2 = d.left + d.width, dy2 = d.top + d.height, da = d.area;
            const P *x1 = px1_.data(), *y1 = py1_.data(), *x2 = px2_.data(), *y2 = py2_.data(), *pa = pa_.data(), *pw = pw_.data();
            P *c = cost.data() + j * np;
            uint8_t *ov = ov_.data();
            for (size_t i = 0; i < np; i++) {
                // fmax / fmin of bbox_iou written as selects (one maxps / minps each): the same value whenever the detection's side
                // is a number, which it always is -- a NaN can only come from a diverged tracker, and then both forms return the
                // detection's coordinate
                const P x_left = x1[i] > dx1 ? x1[i] : dx1, y_top = y1[i] > dy1 ? y1[i] : dy1;
                const P x_right = x2[i] < dx2 ? x2[i] : dx2, y_bottom = y2[i] < dy2 ? y2[i] : dy2;
                const bool none = x_right <= x_left || y_bottom <= y_top;
                const P inter = (x_right - x_left) * (y_bottom - y_top);
                const P uni = da + pa[i] - inter;
                const P iou = none ? 0.f : inter / uni;
                ov[i] |= iou != 0.f;
                c[i] = -iou + pw[i];
            }
        }
        // Trackers that overlap NO detection all carry the same cost row (1 in every column when active, 2 when not), and an
        // edge to one of them never survives the filters below (1 > 1 - iou_threshold; == 2.0).  Of each of the two classes at
        // most n_dets members can take part in an optimal assignment and which ones is immaterial, so only the first n_dets of
        // each class stay in the problem: same optimum value, same surviving edges, and the solver's work drops from
        // n_dets^2 x trackers to about n_dets^3 (at the experiment's maxage 60 / minhits 30 a stream carries 100 - 250
        // trackers, nearly all of them idle: 50 -> 3 us per frame at 11 detections x 256 trackers).
        // Only with iou_threshold > 0: at a threshold <= 0 (the element property allows it) an edge of IoU 0 does pass
        // the filter, the reference then matches and updates idle trackers, and WHICH idle tracker stays in the problem is no
        // longer immaterial -- the full problem is solved.
        std::vector<size_t> &keep = keep_;
        keep.clear();
        size_t idle[2] = {0, 0};
        const bool prune = iou_threshold > (P)0;
        for (size_t i = 0; i < np; i++)
            if (!prune || ov_[i] || idle[trackers[i]->active ? 0 : 1]++ < nd) keep.push_back(i);
        if (keep.size() < np) {
            const size_t nk = keep.size();
            red_.resize(nk * nd);
            for (size_t j = 0; j < nd; j++)
                for (size_t k = 0; k < nk; k++) red_[j * nk + k] = cost[j * np + keep[k]];
            for (auto &e : linear_assignment(red_, nk, nd)) {
                const size_t i = keep[e.first];
                const P thr = trackers[i]->active ? (1.f - iou_threshold) : (2.f - iou_threshold);
                if (cost[e.second * np + i] <= thr) res.emplace_back(i, e.second);
            }
            return res;
        }
        for (auto &e : linear_assignment(cost, np, nd)) {
            const P thr = trackers[e.first]->active ? (1.f - iou_threshold) : (2.f - iou_threshold);
            if (cost[e.second * np + e.first] <= thr) res.push_back(e);
        }
        return res;
    }

    // Sort::update, lib.rs:134-187.  Returns false if a Kalman update failed.
    bool update(std::vector<covahip_bbox> dets, uint64_t pts, std::vector<Tracker> &dead) {
        frame_count += 1;
        const size_t n_dets = dets.size();
        // the arrays match_dets reads (prediction i = prior box of tracker i) follow the trackers: an entry is rewritten when its
        // tracker's prior moves (predict() says so), the retain below compacts them with the trackers
        const size_t nt = trackers.size();
        px1_.resize(nt); py1_.resize(nt); px2_.resize(nt); py2_.resize(nt); pa_.resize(nt); pw_.resize(nt);
        for (size_t i = 0; i < nt; i++) {
            Tracker &t = trackers[i];
            if (t->predict(pts) || soa_stale) {
                const covahip_bbox &b = t->k->prior_box;
                px1_[i] = b.left; py1_[i] = b.top; px2_[i] = b.left + b.width; py2_[i] = b.top + b.height; pa_[i] = b.area;
            }
            pw_[i] = t->active ? 1.f : 2.f;  // lib.rs:108-113
        }
        soa_stale = false;
        auto matches = match_dets(dets);
        // (a tracker / detection appears in at most one pair; the first pair of a tracker wins, as in the reference's `find`)
        match_of_trk.assign(trackers.size(), -1);
        det_matched.assign(n_dets, 0);
        for (auto &e : matches) {
            if (match_of_trk[e.first] < 0) match_of_trk[e.first] = (long)e.second;
            det_matched[e.second] = 1;
        }
        std::vector<size_t> unmatched;
        for (size_t j = 0; j < n_dets; j++)
            if (!det_matched[j]) unmatched.push_back(j);
        for (size_t i = 0; i < trackers.size(); i++) {
            const covahip_bbox *det = nullptr;
            if (match_of_trk[i] >= 0) {
                covahip_bbox &d = dets[(size_t)match_of_trk[i]];
                d.has_timestamp = 1;
                d.timestamp = pts;
                det = &d;
            }
            if (!trackers[i]->update(det)) return false;
        }
        for (auto &t : trackers) t->check_activate(min_hits);
        // retain (lib.rs:166-177) in place: nothing moves in the usual frame in which no tracker dies
        size_t w = 0;
        for (size_t i = 0; i < trackers.size(); i++) {
            Tracker &t = trackers[i];
            if (!t->should_live(max_age)) {
                if (t->active) {
                    t->trim_dead_history();
                    dead.push_back(std::move(t));
                }
            } else {
                if (w != i) {
                    trackers[w] = std::move(t);
                    px1_[w] = px1_[i]; py1_[w] = py1_[i]; px2_[w] = px2_[i]; py2_[w] = py2_[i]; pa_[w] = pa_[i];
                }
                w++;
            }
        }
        trackers.erase(trackers.begin() + (long)w, trackers.end());
        for (size_t j : unmatched) {
            trackers.emplace_back(id_counter, dets[j], pts);
            id_counter += 1;
        }
        return true;
    }

    std::vector<Tracker> finalize() {  // lib.rs:207-213
        std::vector<Tracker> out, keep;
        for (auto &t : trackers) {
            if (t->active) {
                if (t->hist_len() > (size_t)min_hits) out.push_back(std::move(t));
            } else {
                keep.push_back(std::move(t));
            }
        }
        trackers.swap(keep);
        soa_stale = true;
        return out;
    }

    void mark_seen(uint64_t ts) {
        for (auto &t : trackers) t->seen_ts.push_back(ts);
    }
};

int emit_tracks(const std::vector<Tracker> &tracks, covahip_bbox *boxes, size_t cap, size_t *n_boxes,
                uint32_t *track_lens, size_t cap_tracks, size_t *n_tracks) {
    size_t nb = 0;
    bool overflow = false;
    for (size_t k = 0; k < tracks.size(); k++) {
        if (track_lens) {
            if (k < cap_tracks) track_lens[k] = (uint32_t)tracks[k]->hist_len();
            else overflow = true;
        }
        for (const auto &b : tracks[k]->hist()) {
            if (boxes) {
                if (nb < cap) boxes[nb] = b;
                else overflow = true;
            }
            nb++;
        }
    }
    if (n_boxes) *n_boxes = nb;
    if (n_tracks) *n_tracks = tracks.size();
    return overflow ? COVAHIP_ERR_OVERFLOW : COVAHIP_OK;
}

}  // namespace

struct covahip_sort {
    Sort s;
};

extern "C" {

// SSE2 (x86-64 baseline): 8 records per step -- byte minima, three masked shifts per 32-bit record, signed pack to 16 bits.
// (The plain loop is not vectorised by gcc / clang -- stride-4 byte gathers -- and took 7 us per 1080p frame, against 1.9 us
// for the memcpy it replaces; this form takes about 1 us.)
#if defined(__x86_64__)
__attribute__((target("avx2"))) static size_t carrier_pack_avx2(const uint8_t *frame, size_t n_mb, uint16_t *records) {
    const __m256i six = _mm256_set1_epi8(6), m0 = _mm256_set1_epi32(0x7), m1 = _mm256_set1_epi32(0x38), m2 = _mm256_set1_epi32(0x1C0);
    size_t i = 0;
    for (; i + 16 <= n_mb; i += 16) {
        __m256i a = _mm256_min_epu8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(frame + 4 * i)), six);
        __m256i b = _mm256_min_epu8(_mm256_loadu_si256(reinterpret_cast<const __m256i *>(frame + 4 * i + 32)), six);
        a = _mm256_or_si256(_mm256_or_si256(_mm256_and_si256(a, m0), _mm256_and_si256(_mm256_srli_epi32(a, 5), m1)), _mm256_and_si256(_mm256_srli_epi32(a, 10), m2));
        b = _mm256_or_si256(_mm256_or_si256(_mm256_and_si256(b, m0), _mm256_and_si256(_mm256_srli_epi32(b, 5), m1)), _mm256_and_si256(_mm256_srli_epi32(b, 10), m2));
        // packs works per 128-bit half: a.lo b.lo | a.hi b.hi -> restore the order with a 64-bit permute
        _mm256_storeu_si256(reinterpret_cast<__m256i *>(records + i), _mm256_permute4x64_epi64(_mm256_packs_epi32(a, b), 0xD8));
    }
    return i;
}
#endif

void covahip_carrier_pack(const uint8_t *frame, size_t n_mb, uint16_t *records) {
    size_t i = 0;
#if defined(__x86_64__)
    static const bool have_avx2 = __builtin_cpu_supports("avx2");
    if (have_avx2) i = carrier_pack_avx2(frame, n_mb, records);
#endif
#if defined(__SSE2__)
    const __m128i six = _mm_set1_epi8(6), m0 = _mm_set1_epi32(0x7), m1 = _mm_set1_epi32(0x38), m2 = _mm_set1_epi32(0x1C0);
    for (; i + 8 <= n_mb; i += 8) {
        __m128i a = _mm_min_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(frame + 4 * i)), six);
        __m128i b = _mm_min_epu8(_mm_loadu_si128(reinterpret_cast<const __m128i *>(frame + 4 * i + 16)), six);
        a = _mm_or_si128(_mm_or_si128(_mm_and_si128(a, m0), _mm_and_si128(_mm_srli_epi32(a, 5), m1)), _mm_and_si128(_mm_srli_epi32(a, 10), m2));
        b = _mm_or_si128(_mm_or_si128(_mm_and_si128(b, m0), _mm_and_si128(_mm_srli_epi32(b, 5), m1)), _mm_and_si128(_mm_srli_epi32(b, 10), m2));
        _mm_storeu_si128(reinterpret_cast<__m128i *>(records + i), _mm_packs_epi32(a, b));
    }
#endif
    for (; i < n_mb; i++) {
        const unsigned t = frame[4 * i], x = frame[4 * i + 1], y = frame[4 * i + 2];
        records[i] = (uint16_t)((t < 6 ? t : 6) | ((x < 6 ? x : 6) << 3) | ((y < 6 ? y : 6) << 6));
    }
}

int covahip_sort_new(uint64_t max_age, uint64_t min_hits, float iou_threshold, covahip_sort **out) {
    if (!out) return COVAHIP_ERR_INVALID_ARG;
    covahip_sort *p = new (std::nothrow) covahip_sort();
    if (!p) return COVAHIP_ERR_INVALID_ARG;
    p->s.max_age = max_age;
    p->s.min_hits = min_hits;
    p->s.iou_threshold = iou_threshold;
    *out = p;
    return COVAHIP_OK;
}

void covahip_sort_free(covahip_sort *s) { delete s; }

int covahip_sort_update(covahip_sort *s, const covahip_bbox *dets, size_t n_dets, uint64_t pts,
                        covahip_bbox *dead_boxes, size_t cap, size_t *n_dead_boxes, uint32_t *track_lens,
                        size_t cap_tracks, size_t *n_tracks) {
    if (!s || (!dets && n_dets)) return COVAHIP_ERR_INVALID_ARG;
    std::vector<covahip_bbox> d(dets, dets + n_dets);
    std::vector<Tracker> dead;
    if (!s->s.update(std::move(d), pts, dead)) return COVAHIP_ERR_BAD_DATA;
    return emit_tracks(dead, dead_boxes, cap, n_dead_boxes, track_lens, cap_tracks, n_tracks);
}

int covahip_sort_finalize(covahip_sort *s, covahip_bbox *boxes, size_t cap, size_t *n_boxes,
                          uint32_t *track_lens, size_t cap_tracks, size_t *n_tracks) {
    if (!s) return COVAHIP_ERR_INVALID_ARG;
    std::vector<Tracker> fin = s->s.finalize();
    return emit_tracks(fin, boxes, cap, n_boxes, track_lens, cap_tracks, n_tracks);
}

int covahip_sort_mark_seen(covahip_sort *s, uint64_t ts) {
    if (!s) return COVAHIP_ERR_INVALID_ARG;
    s->s.mark_seen(ts);
    return COVAHIP_OK;
}

int covahip_sort_num_trackers(const covahip_sort *s, size_t *n) {
    if (!s || !n) return COVAHIP_ERR_INVALID_ARG;
    *n = s->s.trackers.size();
    return COVAHIP_OK;
}

int covahip_sort_tracker_info(const covahip_sort *s, size_t i, uint64_t *id, int *active,
                              uint64_t *hit_streaks, uint64_t *time_since_update, covahip_bbox *state) {
    if (!s || i >= s->s.trackers.size()) return COVAHIP_ERR_INVALID_ARG;
    const Tracker &t = s->s.trackers[i];
    if (id) *id = t->id;
    if (active) *active = t->active ? 1 : 0;
    if (hit_streaks) *hit_streaks = t->hit_streaks;
    if (time_since_update) *time_since_update = t->time_since_update;
    if (state) *state = from_x(t->k->x);  // get_state(): box of the current estimate
    return COVAHIP_OK;
}

int covahip_sort_tracker_predict(covahip_sort *s, size_t i, uint64_t ts, covahip_bbox *last) {
    if (!s || i >= s->s.trackers.size()) return COVAHIP_ERR_INVALID_ARG;
    s->s.trackers[i]->predict(ts);
    s->s.soa_stale = true;
    if (last) *last = s->s.trackers[i]->hist().back();
    return COVAHIP_OK;
}

int covahip_sort_tracker_update(covahip_sort *s, size_t i, const covahip_bbox *det) {
    if (!s || i >= s->s.trackers.size()) return COVAHIP_ERR_INVALID_ARG;
    s->s.soa_stale = true;
    return s->s.trackers[i]->update(det) ? COVAHIP_OK : COVAHIP_ERR_BAD_DATA;
}

size_t covahip_linear_assignment(const float *cost_colmajor, size_t n_rows, size_t n_cols, uint32_t *pairs,
                                 size_t cap_pairs) {
    if (!cost_colmajor) return 0;
    std::vector<P> c(cost_colmajor, cost_colmajor + n_rows * n_cols);
    auto e = linear_assignment(c, n_rows, n_cols);
    for (size_t k = 0; k < e.size() && k < cap_pairs; k++) {
        pairs[2 * k] = (uint32_t)e[k].first;
        pairs[2 * k + 1] = (uint32_t)e[k].second;
    }
    return e.size();
}

}  // extern "C"

// ============================================================== cova GoP filter
namespace {

struct Au {
    uint64_t id, pts;
    uint32_t flags;
};
struct Gop {  // (min, max, in, out, finalized) -- cova/imp.rs:58-67
    uint64_t min, max;
    std::list<Au> in, out;
    bool finalized;
};

constexpr uint64_t SECOND = 1000000000ull;

}  // namespace

struct covahip_gopfilter {
    covahip_gopfilter_cfg cfg;
    std::list<Gop> bufs;
    // The reference walks ALL buffered GoPs on every mask buffer (imp.rs:135-315).  Its encoded branch runs ahead of the mask branch
    // through an unbounded queue (pipeline.py:237-253), so hundreds of GoPs can be waiting: 23 - 28 % of the chain's CPU samples
    // sat in these walks with 1,200 GoPs buffered per stream.  While the GoPs' pts ranges are in order -- every GoP opened so far
    // started at or behind the largest pts of the one before it, which is what a stream of closed GoPs gives -- a walk may stop at
    // the first GoP that lies wholly on the far side of the range it looks for: the same GoPs are visited in the same order, the
    // rest would have been skipped by the reference's own tests.  `ordered` turns false for good on the first GoP that breaks the
    // rule (and on any pts that lowers a GoP's minimum below its predecessor's maximum); the full walks run from then on.
    bool ordered = true;
    Sort *sort = nullptr;
    bool have_range_start = false;
    uint64_t range_start = 0;
    uint64_t dropped = 0, decoded_dependency = 0, decoded_inference = 0;
    uint32_t next_list = 0;
    std::vector<uint64_t> dropped_ids;   // ids of access units discarded since the last take_dropped
    std::vector<uint8_t> track_wire;     // length-delimited bincode Frames of finished tracks (cova/tracker.rs:59-83)
    ~covahip_gopfilter() { delete sort; }
};

namespace {

struct OutSink {
    covahip_au_out *out;
    size_t cap;
    size_t n = 0;
    void push_list(covahip_gopfilter *g, std::list<Au> &lst) {
        const uint32_t li = g->next_list++;
        for (const Au &a : lst) {
            if (out && n < cap) {
                out[n].id = a.id;
                out[n].pts = a.pts;
                out[n].flags = a.flags;
                out[n].list = li;
            }
            n++;
        }
        lst.clear();
    }
};

// cova/tracker.rs:59-83 / :91-118: every finished track goes out as one length-delimited bincode Frame
// {range_start, oldest = smallest start among the live trackers, history}.  (The reference re-sends the
// frames already in its buffer for every further track of the same call -- not reproduced.)
uint64_t oldest_start(const covahip_gopfilter *g) {   // get_oldest_timestamp (cova/tracker.rs:85-90)
    uint64_t oldest = UINT64_MAX;
    for (const Tracker &t : g->sort->trackers) oldest = std::min(oldest, t->start);
    return oldest;
}
void export_tracks(covahip_gopfilter *g, const std::vector<Tracker> &tracks, uint64_t oldest) {
    if (tracks.empty()) return;
    for (const Tracker &t : tracks) {
        const std::vector<covahip_bbox> &h = t->hist();
        const size_t fl = covahip_frame_serialize(g->range_start, oldest, h.data(), h.size(), nullptr, 0, nullptr);
        const size_t at = g->track_wire.size();
        g->track_wire.resize(at + 4 + fl);
        uint8_t *o = g->track_wire.data() + at;
        o[0] = (uint8_t)(fl >> 24); o[1] = (uint8_t)(fl >> 16); o[2] = (uint8_t)(fl >> 8); o[3] = (uint8_t)fl;
        int st = 0;
        covahip_frame_serialize(g->range_start, oldest, h.data(), h.size(), o + 4, fl, &st);
    }
}

}  // namespace

extern "C" {

void covahip_gopfilter_default_cfg(covahip_gopfilter_cfg *cfg) {
    if (!cfg) return;
    cfg->sort_iou = 0.1f;
    cfg->sort_maxage = 30;
    cfg->sort_minhits = 30;
    cfg->alpha = 0;
    cfg->beta = 0;
    cfg->infer_i = 0;
}

int covahip_gopfilter_new(const covahip_gopfilter_cfg *cfg, covahip_gopfilter **out) {
    if (!cfg || !out) return COVAHIP_ERR_INVALID_ARG;
    covahip_gopfilter *g = new (std::nothrow) covahip_gopfilter();
    if (!g) return COVAHIP_ERR_INVALID_ARG;
    g->cfg = *cfg;
    *out = g;
    return COVAHIP_OK;
}

void covahip_gopfilter_free(covahip_gopfilter *g) { delete g; }

int covahip_gopfilter_push_enc(covahip_gopfilter *g, uint64_t id, uint64_t pts, uint32_t flags) {
    if (!g) return COVAHIP_ERR_INVALID_ARG;
    if (!(flags & COVAHIP_AU_DELTA_UNIT)) {  // imp.rs:327-347: key frame opens a GoP
        if (!g->bufs.empty()) {
            g->bufs.back().finalized = true;
            if (pts < g->bufs.back().max) g->ordered = false;
        }
        Gop gop;
        gop.min = gop.max = pts;
        gop.finalized = false;
        gop.in.push_back(Au{id, pts, flags | COVAHIP_AU_DISCONT});
        g->bufs.push_back(std::move(gop));
    } else {  // imp.rs:348-358
        if (g->bufs.empty()) return COVAHIP_ERR_BAD_DATA;  // reference: unwrap() panic
        Gop &back = g->bufs.back();
        if (pts < back.min) {
            back.min = pts;
            if (g->bufs.size() > 1 && pts < std::prev(g->bufs.end(), 2)->max) g->ordered = false;
        } else if (pts > back.max) back.max = pts;
        back.in.push_back(Au{id, pts, flags});
    }
    return COVAHIP_OK;
}

int covahip_gopfilter_push_boxes(covahip_gopfilter *g, const covahip_bbox *boxes, size_t n, uint64_t pts,
                                 covahip_au_out *out, size_t cap, size_t *n_out) {
    if (!g || (!boxes && n)) return COVAHIP_ERR_INVALID_ARG;
    OutSink sink{out, cap};
    if (!g->sort) {  // imp.rs:99-108 (tracker dims 45x80 are unused by Sort)
        g->sort = new Sort();
        g->sort->max_age = g->cfg.sort_maxage;
        g->sort->min_hits = g->cfg.sort_minhits;
        g->sort->iou_threshold = (P)(double)g->cfg.sort_iou;
    }
    // cova/tracker.rs:43-60
    if (!g->have_range_start) {
        g->have_range_start = true;
        g->range_start = pts;
    }
    std::vector<Tracker> dead;
    if (!g->sort->update(std::vector<covahip_bbox>(boxes, boxes + n), pts, dead)) return COVAHIP_ERR_BAD_DATA;
    if (!dead.empty()) export_tracks(g, dead, oldest_start(g));   // after the update, over the trackers that are left (tracker.rs:51-58)
    bool have_min = !dead.empty();
    uint64_t min_track_pts = 0;
    for (const Tracker &t : dead)
        if (!t->is_seen()) min_track_pts = std::max(min_track_pts, t->start);

    const uint64_t clk30 = SECOND / 30;
    const uint64_t maxage_pts = clk30 * ((uint64_t)g->cfg.sort_maxage + 10);  // SAFETY_BUFFER = 10
    const uint64_t max_track_pts = pts >= maxage_pts ? pts - maxage_pts : 0;

    if (have_min) {  // imp.rs:135-253
        size_t track_inferenced = 0;
        uint64_t dd = 0, di = 0;
        for (auto it = g->bufs.rbegin(); it != g->bufs.rend(); ++it) {
            Gop &gop = *it;
            if (g->ordered && gop.max < min_track_pts) break;   // newest first: every older GoP ends even earlier
            if (!(min_track_pts <= gop.max && gop.min <= max_track_pts)) continue;
            bool already = false;
            for (const Au &a : gop.out)
                if (min_track_pts < a.pts) {
                    track_inferenced += 1;
                    already = true;
                    break;
                }
            if (already) continue;
            while (!gop.in.empty()) {
                Au buf = gop.in.front();
                gop.in.pop_front();
                if (track_inferenced > 0) {  // NB: the popped AU is discarded (reference behaviour)
                    g->dropped_ids.push_back(buf.id);
                    break;
                }
                if (min_track_pts <= buf.pts) {
                    g->sort->mark_seen(buf.pts);
                    di += 1;
                    gop.out.push_back(buf);
                    track_inferenced += 1;
                    break;
                } else {
                    buf.flags |= COVAHIP_AU_DROPPABLE;
                    dd += 1;
                    gop.out.push_back(buf);
                }
            }
        }
        if (track_inferenced < (size_t)g->cfg.beta) {  // imp.rs:200-246
            for (auto it = g->bufs.rbegin(); it != g->bufs.rend(); ++it) {
                Gop &gop = *it;
                if (g->ordered && gop.max < min_track_pts) break;
                if (!(min_track_pts <= gop.max && gop.min <= max_track_pts)) continue;
                if (gop.out.empty()) continue;
                const size_t extra_decode = std::min(gop.in.size(), (size_t)g->cfg.alpha);
                const size_t extra_infer = std::min(extra_decode, (size_t)g->cfg.beta - track_inferenced);
                if (extra_decode == 0 || extra_infer == 0) continue;
                const size_t step = extra_decode / extra_infer, rem = extra_decode % extra_infer;
                for (size_t k = 0; k < rem; k++) {
                    Au b = gop.in.front();
                    gop.in.pop_front();
                    b.flags |= COVAHIP_AU_DROPPABLE;
                    dd += 1;
                    gop.out.push_back(b);
                }
                for (size_t k = 0; k < extra_infer; k++) {
                    const size_t ndep = step > 0 ? step - 1 : 0;
                    for (size_t q = 0; q < ndep; q++) {
                        Au b = gop.in.front();
                        gop.in.pop_front();
                        b.flags |= COVAHIP_AU_DROPPABLE;
                        dd += 1;
                        gop.out.push_back(b);
                    }
                    Au b = gop.in.front();
                    gop.in.pop_front();
                    g->sort->mark_seen(b.pts);
                    di += 1;
                    gop.out.push_back(b);
                    track_inferenced += 1;
                }
            }
        }
        if (track_inferenced == 0) {  // reference: assert!(track_inferenced > 0) -> panic -> FlowError
            if (n_out) *n_out = 0;
            return COVAHIP_ERR_BAD_DATA;
        }
        g->decoded_inference += di;
        g->decoded_dependency += dd;
    }

    // imp.rs:255-315: flush finalised GoPs older than 250 frames
    uint64_t dropped = 0, di2 = 0;
    const uint64_t gop_pts = clk30 * 250;
    const uint64_t droppable_pts = pts >= gop_pts ? pts - gop_pts : 0;
    for (auto it = g->bufs.begin(); it != g->bufs.end();) {
        Gop &gop = *it;
        if (g->ordered && gop.max > droppable_pts) break;   // oldest first: every newer GoP ends even later
        if (!(gop.finalized && gop.max <= droppable_pts)) {
            ++it;
            continue;
        }
        if (g->cfg.infer_i && !gop.in.empty()) {
            Au b = gop.in.front();
            gop.in.pop_front();
            if (!(b.flags & COVAHIP_AU_DELTA_UNIT)) {
                di2 += 1;
                gop.out.push_back(b);
            } else {
                dropped += 1;
                g->dropped_ids.push_back(b.id);
            }
        }
        if (!gop.out.empty()) sink.push_list(g, gop.out);
        dropped += gop.in.size();
        for (const Au &a : gop.in) g->dropped_ids.push_back(a.id);
        it = g->bufs.erase(it);
    }
    g->decoded_inference += di2;
    g->dropped += dropped;
    if (n_out) *n_out = sink.n;
    return (out && sink.n > cap) ? COVAHIP_ERR_OVERFLOW : COVAHIP_OK;
}

int covahip_gopfilter_eos(covahip_gopfilter *g, covahip_au_out *out, size_t cap, size_t *n_out) {
    if (!g) return COVAHIP_ERR_INVALID_ARG;
    OutSink sink{out, cap};
    uint64_t dropped = 0;
    for (Gop &gop : g->bufs) {  // imp.rs:371-387
        dropped += gop.in.size();
        for (const Au &a : gop.in) g->dropped_ids.push_back(a.id);
        sink.push_list(g, gop.out);
    }
    g->bufs.clear();
    g->dropped += dropped;
    if (g->sort) {   // tracker.take().flush(): `oldest` over ALL live trackers, taken BEFORE finalize() removes the active ones (cova/tracker.rs:97-99)
        const uint64_t oldest = oldest_start(g);
        export_tracks(g, g->sort->finalize(), oldest);
    }
    delete g->sort;
    g->sort = nullptr;
    if (n_out) *n_out = sink.n;
    return (out && sink.n > cap) ? COVAHIP_ERR_OVERFLOW : COVAHIP_OK;
}

int covahip_gopfilter_take_dropped(covahip_gopfilter *g, uint64_t *ids, size_t cap, size_t *n) {
    if (!g || !n || (!ids && cap)) return COVAHIP_ERR_INVALID_ARG;
    const size_t k = std::min(cap, g->dropped_ids.size());
    std::copy(g->dropped_ids.begin(), g->dropped_ids.begin() + k, ids);
    g->dropped_ids.erase(g->dropped_ids.begin(), g->dropped_ids.begin() + k);
    *n = k;
    return COVAHIP_OK;
}

size_t covahip_gopfilter_take_track_export(covahip_gopfilter *g, uint8_t *out, size_t cap, int *status) {
    if (!g) {
        if (status) *status = COVAHIP_ERR_INVALID_ARG;
        return 0;
    }
    const size_t need = g->track_wire.size();
    if (out && need <= cap) {
        std::copy(g->track_wire.begin(), g->track_wire.end(), out);
        g->track_wire.clear();
        if (status) *status = COVAHIP_OK;
    } else if (status) {
        *status = COVAHIP_ERR_OVERFLOW;
    }
    return need;
}

int covahip_gopfilter_counters(const covahip_gopfilter *g, uint64_t *dropped, uint64_t *decoded_dependency,
                               uint64_t *decoded_inference) {
    if (!g) return COVAHIP_ERR_INVALID_ARG;
    if (dropped) *dropped = g->dropped;
    if (decoded_dependency) *decoded_dependency = g->decoded_dependency;
    if (decoded_inference) *decoded_inference = g->decoded_inference;
    return COVAHIP_OK;
}

}  // extern "C"

// ============================================================== sink formats + track export
// (data formats either side of the hot path)
//   tfrecordsink  cova-rs/gst-plugins/src/tfrecordsink/imp.rs:69-198  one tf.train.Example per frame / GoP
//   bboxsink      cova-rs/gst-plugins/src/bboxsink/imp.rs:252-270     serde CSV rows of Bbox
//   track export  cova-rs/gst-plugins/src/cova/tracker.rs:59-83       LengthDelimitedCodec(bincode Frame)
#include <charconv>
#include <string>

namespace {

uint32_t crc32c_table[256];
bool crc32c_ready = false;
uint32_t crc32c(const uint8_t *p, size_t n) {  // Castagnoli, reflected, as TFRecord uses it
    if (!crc32c_ready) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
            crc32c_table[i] = c;
        }
        crc32c_ready = true;
    }
    uint32_t c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; i++) c = crc32c_table[(c ^ p[i]) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}
uint32_t masked_crc(const uint8_t *p, size_t n) {
    const uint32_t c = crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}
void put_varint(std::string &s, uint64_t v) {
    while (v >= 0x80) { s.push_back((char)(v | 0x80)); v >>= 7; }
    s.push_back((char)v);
}
void put_len_field(std::string &s, int field, const std::string &payload) {
    put_varint(s, (uint64_t)(field << 3 | 2));
    put_varint(s, payload.size());
    s += payload;
}

// serde/csv formats f32 with ryu (crate ryu, pretty::format32): shortest round-trip decimal digits
// d1..dn with decimal exponent k (value = d1..dn x 10^k), kk = n + k, then
//   0 <= k  and kk <= 13 : digits, k zeros, ".0"          1234e7  -> 12340000000.0
//   0 <  kk and kk <= 13 : decimal point inside            1234e-2 -> 12.34
//   -6 < kk and kk <= 0  : "0." + (-kk) zeros + digits      1234e-6 -> 0.001234
//   otherwise            : d1[.d2..dn]e(kk-1)                1e21, 1.234e33, 1e-7
std::string ryu_like(float v) {
    if (std::isnan(v)) return "NaN";
    if (std::isinf(v)) return v > 0 ? "inf" : "-inf";
    if (v == 0.f) return std::signbit(v) ? "-0.0" : "0.0";
    char buf[64];
    auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);   // shortest digits
    std::string s(buf, r.ptr);
    std::string sign;
    if (s[0] == '-') { sign = "-"; s = s.substr(1); }
    const size_t e = s.find('e');
    std::string digits;
    for (char c : s.substr(0, e)) if (c != '.') digits += c;
    const int sci = std::atoi(s.c_str() + e + 1);
    const int n = (int)digits.size(), kk = sci + 1, k = kk - n;
    std::string out;
    if (k >= 0 && kk <= 13) out = digits + std::string(k, '0') + ".0";
    else if (kk > 0 && kk <= 13) out = digits.substr(0, kk) + "." + digits.substr(kk);
    else if (kk > -6 && kk <= 0) out = "0." + std::string(-kk, '0') + digits;
    else out = digits.substr(0, 1) + (n > 1 ? "." + digits.substr(1) : "") + "e" + std::to_string(kk - 1);
    return sign + out;
}

}  // namespace

extern "C" {

// One framed TFRecord record holding a tf.train.Example with the four bytes_list features
// mb_type / mv_x / mv_y / gt, one w*h-byte string per frame (tfrecordsink/imp.rs:105-130), zero
// filled up to `pad_to_frames` strings when that is larger than n_frames (the `gop` property,
// imp.rs:156-167).  rgba: [n_frames][h][w][4]; gt: [n_frames][h*w].  Feature order is fixed here
// (the reference's HashMap order is random): parity is at parsed-record level.
size_t covahip_tfrecord_example(const uint8_t *rgba, const uint8_t *gt, int n_frames, int pad_to_frames, int w, int h,
                                uint8_t *out, size_t cap, int *status) {
    const size_t hw = (size_t)w * h;
    const int total = pad_to_frames > n_frames ? pad_to_frames : n_frames;
    static const char *names[4] = {"mb_type", "mv_x", "mv_y", "gt"};
    std::string features;
    for (int f = 0; f < 4; f++) {
        std::string blist;
        for (int i = 0; i < total; i++) {
            std::string v(hw, '\0');
            if (i < n_frames) {
                if (f < 3) for (size_t p = 0; p < hw; p++) v[p] = (char)rgba[((size_t)i * hw + p) * 4 + f];
                else if (gt) std::memcpy(&v[0], gt + (size_t)i * hw, hw);
            }
            put_len_field(blist, 1, v);                       // BytesList.value
        }
        std::string feature;
        put_len_field(feature, 1, blist);                     // Feature.bytes_list
        std::string entry;
        put_len_field(entry, 1, names[f]);                    // map key
        put_len_field(entry, 2, feature);                     // map value
        put_len_field(features, 1, entry);                    // Features.feature
    }
    std::string example;
    put_len_field(example, 1, features);                      // Example.features
    const uint64_t len = example.size();
    const size_t need = 8 + 4 + example.size() + 4;
    if (status) *status = (out && need <= cap) ? COVAHIP_OK : COVAHIP_ERR_OVERFLOW;
    if (out && need <= cap) {
        std::memcpy(out, &len, 8);
        const uint32_t c1 = masked_crc(out, 8);
        std::memcpy(out + 8, &c1, 4);
        std::memcpy(out + 12, example.data(), example.size());
        const uint32_t c2 = masked_crc((const uint8_t *)example.data(), example.size());
        std::memcpy(out + 12 + example.size(), &c2, 4);
    }
    return need;
}

// CSV text of bboxsink: header (serde field order of Bbox, bbox.rs:4-14) and one row per box;
// None -> empty field.
size_t covahip_bbox_csv(const covahip_bbox *boxes, size_t n, int with_header, char *out, size_t cap, int *status) {
    std::string s;
    if (with_header) s += "left,top,width,height,area,track_id,timestamp,class_id,confidence\n";
    for (size_t i = 0; i < n; i++) {
        const covahip_bbox &b = boxes[i];
        s += ryu_like(b.left) + "," + ryu_like(b.top) + "," + ryu_like(b.width) + "," + ryu_like(b.height) + "," +
             ryu_like(b.area) + ",";
        if (b.has_track_id) s += std::to_string(b.track_id);
        s += ",";
        if (b.has_timestamp) s += std::to_string(b.timestamp);
        s += ",";
        if (b.has_class_id) s += std::to_string(b.class_id);
        s += ",";
        if (b.has_confidence) s += ryu_like(b.confidence);
        s += "\n";
    }
    if (status) *status = (out && s.size() <= cap) ? COVAHIP_OK : COVAHIP_ERR_OVERFLOW;
    if (out && s.size() <= cap) std::memcpy(out, s.data(), s.size());
    return s.size();
}

// Track export of cova::Tracker (cova/tracker.rs:59-83): for every dead track one
// LengthDelimitedCodec frame = 4-byte big-endian length + bincode(Frame{range_start, oldest,
// bboxes = the track's history}).  tracks: flattened histories + per-track lengths as
// covahip_sort_update returns them.  (The reference re-sends earlier frames because its buffer is
// never cleared; that bug is not reproduced: each track is emitted once.)
size_t covahip_tracks_export(uint64_t range_start, uint64_t oldest, const covahip_bbox *boxes, const uint32_t *track_lens,
                             size_t n_tracks, uint8_t *out, size_t cap, int *status) {
    size_t need = 0, off = 0;
    for (size_t t = 0; t < n_tracks; t++) {
        const size_t fl = covahip_frame_serialize(range_start, oldest, boxes + off, track_lens[t], nullptr, 0, nullptr);
        if (out && need + 4 + fl <= cap) {
            const uint32_t be = (uint32_t)fl;
            out[need] = (uint8_t)(be >> 24); out[need + 1] = (uint8_t)(be >> 16);
            out[need + 2] = (uint8_t)(be >> 8); out[need + 3] = (uint8_t)be;
            int st = 0;
            covahip_frame_serialize(range_start, oldest, boxes + off, track_lens[t], out + need + 4, fl, &st);
        }
        need += 4 + fl;
        off += track_lens[t];
    }
    if (status) *status = (out && need <= cap) ? COVAHIP_OK : COVAHIP_ERR_OVERFLOW;
    return need;
}

}  // extern "C"

// ===================================================================== analysis-aggregator join
// Association of tracker output with DNN detections, restating
// cova-rs/analysis-aggregator/src/server/assoc.rs (Associator :63-441, message loop :443-507),
// track.rs:59-66 and dnn.rs:57-86.  The reference visits three HashMaps; the orders are fixed
// here: classes in ascending id, the LAST of equally frequent classes is "the most frequent" (what
// max_by_key returns over an ascending visit), ranges ascending at terminate.  Quirks kept: `>` vs `>=` moving_iou for a
// new vs an existing track, the always-true "two detections" filter, tracks still pending at
// terminate never reach assoc.csv.
namespace {

struct StationaryObj {   // assoc.rs:11-58
    uint64_t range_start, range_end, start, end;
    covahip_bbox bbox;
    uint32_t class_id;
};

void bbox_scale(covahip_bbox &b, float s) {   // bbox.rs:69-82: the centroid stays
    if (s == 1.f) return;
    const float x = b.left + b.width / 2.f, y = b.top + b.height / 2.f;
    b.width *= s;
    b.height *= s;
    b.left = x - b.width / 2.f;
    b.top = y - b.height / 2.f;
    b.area *= s * s;
}
void bbox_scale_dim(covahip_bbox &b, float s) {   // bbox.rs:58-67
    if (s == 1.f) return;
    b.left *= s; b.top *= s; b.width *= s; b.height *= s;
    b.area *= s * s;
}

std::string csv_rows(const std::vector<covahip_bbox> &rows) {
    if (rows.empty()) return std::string();   // csv::Writer emits the header with the first record
    const size_t n = covahip_bbox_csv(rows.data(), rows.size(), 1, nullptr, 0, nullptr);
    std::string out(n, '\0');
    int st = 0;
    covahip_bbox_csv(rows.data(), rows.size(), 1, &out[0], n, &st);
    return out;
}

}  // namespace

struct covahip_assoc {
    struct Track { uint64_t range_start, range_end; std::vector<covahip_bbox> boxes; };
    struct Dnn { bool matched; covahip_bbox box; };
    std::map<uint64_t, uint64_t> tracker_range;
    std::vector<covahip_bbox> rows[4];   // track, dnn, assoc, stationary
    std::list<Track> tracks;
    std::list<Dnn> dnns;
    std::list<StationaryObj> stationary, finalized;
    std::map<uint64_t, std::vector<uint32_t>> track2class;
    float moving_iou = 0.15f, stationary_iou = 0.3f, scale_factor = 1.3f;
    uint64_t stationary_maxage = 120ull * 1000000000ull, max_track_id = 0;
    std::string dnn_text;                // unparsed tail of the detection text stream
    bool terminated = false;

    void finalize_trk(uint64_t ts) {     // assoc.rs:127-215
        for (auto it = tracks.begin(); it != tracks.end();) {
            if (!(it->range_start <= ts && ts < it->range_end && it->boxes.back().timestamp < ts)) { ++it; continue; }
            std::vector<uint32_t> class_ids;
            auto f = track2class.find(it->boxes.front().track_id);
            if (f != track2class.end() && !f->second.empty()) {
                std::map<uint32_t, int> count;
                for (uint32_t c : f->second) count[c]++;
                uint32_t best = 0;
                int freq = -1;
                for (auto &kv : count) if (kv.second >= freq) { best = kv.first; freq = kv.second; }
                count.erase(best);
                class_ids.push_back(best);
                for (auto &kv : count) if (freq == 1 || kv.second >= 2) class_ids.push_back(kv.first);
            }
            if (f != track2class.end()) track2class.erase(f);
            for (uint32_t c : class_ids)
                for (auto &b : it->boxes) { b.class_id = c; b.has_class_id = 1; rows[2].push_back(b); }
            it = tracks.erase(it);
        }
    }
    void finalize_dnn(uint64_t rs, uint64_t re, uint64_t ts) {   // assoc.rs:220-267
        for (auto it = dnns.begin(); it != dnns.end();) {
            const uint64_t dt = it->box.timestamp;
            if (!(rs <= dt && dt < re && dt < ts)) { ++it; continue; }
            if (!it->matched) {
                StationaryObj *best = nullptr;
                float best_iou = 0.f;
                for (auto &so : stationary) {
                    if (so.range_start != rs || so.class_id != it->box.class_id) continue;
                    const float iou = bbox_iou(so.bbox, it->box);
                    if (iou >= stationary_iou && (!best || iou >= best_iou)) { best = &so; best_iou = iou; }
                }
                if (best) best->end = dt;
                else stationary.push_back(StationaryObj{rs, re, dt, dt, it->box, it->box.class_id});
            }
            it = dnns.erase(it);
        }
    }
    void finalize_stationary(uint64_t ts) {   // assoc.rs:271-287
        for (auto it = stationary.begin(); it != stationary.end();) {
            if (it->range_start <= ts && ts < it->range_end && stationary_maxage + it->end < ts) {
                if (it->range_start != it->range_end) finalized.push_back(*it);   // the reference's filter, always true
                it = stationary.erase(it);
            } else ++it;
        }
    }
    // the track's box at `ts`, scaled about its centre (assoc.rs:326-334, 396-404); false = the reference unwraps None
    bool match_box(const std::vector<covahip_bbox> &trk, uint64_t ts, covahip_bbox &out) const {
        for (auto &b : trk)
            if (b.timestamp == ts) { out = b; bbox_scale(out, scale_factor); return true; }
        return false;
    }
    int update_dnn(const covahip_bbox *boxes, size_t n) {   // assoc.rs:296-367
        std::vector<uint64_t> seen;
        for (size_t i = 0; i < n; i++) {
            if (!boxes[i].has_timestamp || !boxes[i].has_class_id) return COVAHIP_ERR_BAD_DATA;
            if (std::find(seen.begin(), seen.end(), boxes[i].timestamp) == seen.end()) seen.push_back(boxes[i].timestamp);
        }
        for (uint64_t ts : seen) { finalize_stationary(ts); finalize_trk(ts); }
        for (size_t i = 0; i < n; i++) {
            const covahip_bbox &d = boxes[i];
            rows[1].push_back(d);
            bool matched = false;
            for (auto &t : tracks) {
                if (!(t.range_start <= d.timestamp && d.timestamp < t.range_end && t.boxes.front().timestamp <= d.timestamp)) continue;
                covahip_bbox tb;
                if (!match_box(t.boxes, d.timestamp, tb)) return COVAHIP_ERR_BAD_DATA;
                if (bbox_iou(tb, d) >= moving_iou) { track2class[tb.track_id].push_back(d.class_id); matched = true; }
            }
            dnns.push_back(Dnn{matched, d});
        }
        return COVAHIP_OK;
    }
    int update_track(uint64_t rs, uint64_t oldest, const covahip_bbox *boxes, size_t n) {   // assoc.rs:370-431
        auto r = tracker_range.find(rs);
        if (r == tracker_range.end() || n == 0) return COVAHIP_ERR_BAD_DATA;
        for (size_t i = 0; i < n; i++) if (!boxes[i].has_timestamp || !boxes[i].has_track_id) return COVAHIP_ERR_BAD_DATA;
        Track t{rs, r->second, std::vector<covahip_bbox>(boxes, boxes + n)};
        for (auto &b : t.boxes) rows[0].push_back(b);
        max_track_id = std::max(max_track_id, t.boxes.front().track_id);
        const uint64_t t0 = t.boxes.front().timestamp, t1 = t.boxes.back().timestamp;
        for (auto &d : dnns) {
            if (!(t0 <= d.box.timestamp && d.box.timestamp <= t1)) continue;
            covahip_bbox tb;
            if (!match_box(t.boxes, d.box.timestamp, tb)) return COVAHIP_ERR_BAD_DATA;
            if (bbox_iou(tb, d.box) > moving_iou) { track2class[tb.track_id].push_back(d.box.class_id); d.matched = true; }
        }
        tracks.push_back(std::move(t));
        finalize_dnn(rs, r->second, oldest);
        return COVAHIP_OK;
    }
    void terminate() {   // assoc.rs:434-467
        if (terminated) return;
        terminated = true;
        for (auto &kv : tracker_range) {
            finalize_trk(kv.second);
            finalize_dnn(kv.first, kv.second, kv.second);
            finalize_stationary(kv.second);
        }
        uint64_t tid = max_track_id + 1;
        for (auto &so : finalized) {
            for (uint64_t ts = so.start; ts < so.end; ts += 100000000ull)   // Stationary::to_vec, assoc.rs:41-57
                for (uint64_t i = 0; i < 2; i++) {
                    covahip_bbox b = so.bbox;
                    b.timestamp = ts + i * 33333333ull; b.has_timestamp = 1;
                    b.track_id = tid; b.has_track_id = 1;
                    rows[3].push_back(b);
                }
            tid++;
        }
    }
};

extern "C" {

void covahip_assoc_default_cfg(covahip_assoc_cfg *cfg) {   // main.rs:32-39
    if (!cfg) return;
    cfg->moving_iou = 0.15f;
    cfg->stationary_iou = 0.3f;
    cfg->stationary_maxage_s = 120;
    cfg->scale_factor = 1.3f;
}

int covahip_assoc_new(const covahip_assoc_cfg *cfg, const uint64_t *range_starts, size_t n_trackers, covahip_assoc **out) {
    if (!cfg || !range_starts || !n_trackers || !out) return COVAHIP_ERR_INVALID_ARG;
    covahip_assoc *a = new covahip_assoc();
    a->moving_iou = cfg->moving_iou;
    a->stationary_iou = cfg->stationary_iou;
    a->stationary_maxage = cfg->stationary_maxage_s * 1000000000ull;
    a->scale_factor = cfg->scale_factor;
    std::vector<uint64_t> rs(range_starts, range_starts + n_trackers);   // assoc.rs:473-489
    std::sort(rs.begin(), rs.end());
    rs.push_back(UINT64_MAX);
    for (size_t i = 0; i < n_trackers; i++) a->tracker_range[rs[i]] = rs[i + 1];
    *out = a;
    return COVAHIP_OK;
}
void covahip_assoc_free(covahip_assoc *a) { delete a; }

int covahip_assoc_push_track(covahip_assoc *a, uint64_t range_start, uint64_t oldest, const covahip_bbox *boxes, size_t n) {
    if (!a || (!boxes && n) || a->terminated) return COVAHIP_ERR_INVALID_ARG;
    return a->update_track(range_start, oldest, boxes, n);
}

// One LengthDelimitedCodec payload as a tracker connection receives it (track.rs:47-66): bincode Frame,
// boxes scaled from macroblocks to pixels (x16), track ids re-based by range_start.
int covahip_assoc_push_track_frame(covahip_assoc *a, const uint8_t *payload, size_t len) {
    if (!a || !payload || a->terminated) return COVAHIP_ERR_INVALID_ARG;
    Reader r{payload, len};
    const uint64_t rs = r.val<uint64_t>(), oldest = r.val<uint64_t>(), cnt = r.val<uint64_t>();
    if (!r.ok || cnt > len) return COVAHIP_ERR_BAD_DATA;
    std::vector<covahip_bbox> boxes((size_t)cnt);
    for (auto &b : boxes) {
        if (!read_bbox(r, b) || !b.has_track_id) return COVAHIP_ERR_BAD_DATA;
        bbox_scale_dim(b, 16.f);
        b.track_id += rs;
    }
    if (r.n != len) return COVAHIP_ERR_BAD_DATA;
    return a->update_track(rs, oldest, boxes.data(), boxes.size());
}

int covahip_assoc_push_dnn(covahip_assoc *a, const covahip_bbox *boxes, size_t n) {
    if (!a || (!boxes && n) || a->terminated) return COVAHIP_ERR_INVALID_ARG;
    return a->update_dnn(boxes, n);
}

// Detection text as it arrives on a DNN connection (dnn.rs:57-86): rows "timestamp,left,top,width,height,class_id\n";
// an incomplete last row is kept for the next call; the complete rows of one call form one update.
int covahip_assoc_push_dnn_text(covahip_assoc *a, const char *text, size_t len) {
    if (!a || (!text && len) || a->terminated) return COVAHIP_ERR_INVALID_ARG;
    a->dnn_text.append(text, len);
    std::vector<covahip_bbox> boxes;
    size_t pos = 0;
    while (true) {
        const size_t nl = a->dnn_text.find('\n', pos);
        if (nl == std::string::npos) break;
        const std::string line = a->dnn_text.substr(pos, nl - pos);
        pos = nl + 1;
        if (line.empty()) continue;
        std::vector<std::string> f;
        size_t p = 0;
        while (true) {
            const size_t c = line.find(',', p);
            f.push_back(line.substr(p, c == std::string::npos ? std::string::npos : c - p));
            if (c == std::string::npos) break;
            p = c + 1;
        }
        if (f.size() != 6 || f[5].empty()) return COVAHIP_ERR_BAD_DATA;
        char *end = nullptr;
        covahip_bbox b = bbox_new(std::strtof(f[1].c_str(), nullptr), std::strtof(f[2].c_str(), nullptr),
                                  std::strtof(f[3].c_str(), nullptr), std::strtof(f[4].c_str(), nullptr));
        b.timestamp = std::strtoull(f[0].c_str(), &end, 10);
        if (*end) return COVAHIP_ERR_BAD_DATA;
        const long long cls = std::strtoll(f[5].c_str(), &end, 10);
        if (*end || cls < 0) return COVAHIP_ERR_BAD_DATA;   // dnn.rs:78: u32::try_from(i32)
        b.class_id = (uint32_t)cls;
        b.has_timestamp = b.has_class_id = 1;
        boxes.push_back(b);
    }
    a->dnn_text.erase(0, pos);
    return boxes.empty() ? COVAHIP_OK : a->update_dnn(boxes.data(), boxes.size());
}

int covahip_assoc_terminate(covahip_assoc *a) {
    if (!a) return COVAHIP_ERR_INVALID_ARG;
    a->terminate();
    return COVAHIP_OK;
}

// CSV text of one of the four output files (0 track.csv, 1 dnn.csv, 2 assoc.csv, 3 stationary.csv) as written so far.
size_t covahip_assoc_csv(covahip_assoc *a, int which, char *out, size_t cap, int *status) {
    if (!a || which < 0 || which > 3) { if (status) *status = COVAHIP_ERR_INVALID_ARG; return 0; }
    const std::string s = csv_rows(a->rows[which]);
    if (status) *status = (out && s.size() <= cap) ? COVAHIP_OK : COVAHIP_ERR_OVERFLOW;
    if (out && s.size() <= cap) std::memcpy(out, s.data(), s.size());
    return s.size();
}

}  // extern "C"
