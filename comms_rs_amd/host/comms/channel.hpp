// channel.hpp -- MPMC channel with crossbeam-channel's disconnect semantics.
//
// The reference moves every message through crossbeam channels
// (src/prelude.rs:5): `unbounded()` for connect_nodes! (src/node/mod.rs:152),
// `bounded(n)` for Graph::new(Some(n)) (src/node/graph.rs:44-47).  What the
// node loop relies on, and what is kept here:
//   * recv() blocks; it fails only when the queue is empty AND every Sender is
//     gone  (-> NodeError::DataEnd, node_derive/src/lib.rs:203);
//   * send() fails when every Receiver is gone (-> NodeError::CommError, :158);
//     on a bounded channel it blocks while the queue is full;
//   * Sender and Receiver are cloneable handles; dropping the last one of a side
//     disconnects the channel and wakes the other side.
#pragma once

#include <condition_variable>
#include <cstddef>
#include <deque>
#include <memory>
#include <mutex>
#include <optional>
#include <utility>
#include <vector>

namespace comms {
namespace channel {

template <class T>
struct Shared {
    std::mutex m;
    std::condition_variable not_empty, not_full;
    std::deque<T> q;
    size_t capacity = 0;  // 0 = unbounded
    size_t senders = 0, receivers = 0;
};

template <class T>
class Sender {
public:
    Sender() = default;
    explicit Sender(std::shared_ptr<Shared<T>> s) : s_(std::move(s)) { attach(); }
    Sender(const Sender& o) : s_(o.s_) { attach(); }
    Sender(Sender&& o) noexcept : s_(std::move(o.s_)) {}
    Sender& operator=(Sender o) noexcept {
        std::swap(s_, o.s_);
        return *this;
    }
    ~Sender() { detach(); }

    // false = every receiver has been dropped (crossbeam SendError)
    bool send(T v) const {
        std::unique_lock<std::mutex> lk(s_->m);
        if (s_->capacity)
            s_->not_full.wait(lk, [&] { return s_->q.size() < s_->capacity || s_->receivers == 0; });
        if (s_->receivers == 0) return false;
        s_->q.push_back(std::move(v));
        lk.unlock();
        s_->not_empty.notify_one();
        return true;
    }

    // The same as send() on every element in order, under one lock while capacity allows
    // (per-sample nodes that process a drained block hand their outputs back this way).
    bool send_many(std::vector<T>&& vs) const {
        size_t i = 0;
        while (i < vs.size()) {
            std::unique_lock<std::mutex> lk(s_->m);
            if (s_->capacity)
                s_->not_full.wait(lk, [&] { return s_->q.size() < s_->capacity || s_->receivers == 0; });
            if (s_->receivers == 0) return false;
            const size_t room = s_->capacity ? s_->capacity - s_->q.size() : vs.size() - i;
            const size_t take = room < vs.size() - i ? room : vs.size() - i;
            for (size_t k = 0; k < take; ++k) s_->q.push_back(std::move(vs[i + k]));
            i += take;
            lk.unlock();
            s_->not_empty.notify_all();
        }
        return true;
    }

private:
    void attach() {
        if (!s_) return;
        std::lock_guard<std::mutex> lk(s_->m);
        ++s_->senders;
    }
    void detach() {
        if (!s_) return;
        bool last;
        {
            std::lock_guard<std::mutex> lk(s_->m);
            last = --s_->senders == 0;
        }
        if (last) s_->not_empty.notify_all();
        s_.reset();
    }
    std::shared_ptr<Shared<T>> s_;
};

template <class T>
class Receiver {
public:
    using value_type = T;
    Receiver() = default;
    explicit Receiver(std::shared_ptr<Shared<T>> s) : s_(std::move(s)) { attach(); }
    Receiver(const Receiver& o) : s_(o.s_) { attach(); }
    Receiver(Receiver&& o) noexcept : s_(std::move(o.s_)) {}
    Receiver& operator=(Receiver o) noexcept {
        std::swap(s_, o.s_);
        return *this;
    }
    ~Receiver() { detach(); }

    // nullopt = disconnected and drained (crossbeam RecvError)
    std::optional<T> recv() const {
        std::unique_lock<std::mutex> lk(s_->m);
        s_->not_empty.wait(lk, [&] { return !s_->q.empty() || s_->senders == 0; });
        if (s_->q.empty()) return std::nullopt;
        T v = std::move(s_->q.front());
        s_->q.pop_front();
        lk.unlock();
        s_->not_full.notify_one();
        return v;
    }
    // non-blocking; nullopt if nothing is queued right now
    std::optional<T> try_recv() const {
        std::lock_guard<std::mutex> lk(s_->m);
        if (s_->q.empty()) return std::nullopt;
        T v = std::move(s_->q.front());
        s_->q.pop_front();
        s_->not_full.notify_one();
        return v;
    }
    // recv() for the first message, then try_recv() for whatever else is already queued (at most
    // `max` in all), appended to `out` in order.  0 = disconnected and drained.
    size_t recv_many(std::vector<T>& out, size_t max) const {
        std::unique_lock<std::mutex> lk(s_->m);
        s_->not_empty.wait(lk, [&] { return !s_->q.empty() || s_->senders == 0; });
        size_t n = 0;
        while (n < max && !s_->q.empty()) {
            out.push_back(std::move(s_->q.front()));
            s_->q.pop_front();
            ++n;
        }
        lk.unlock();
        if (n) s_->not_full.notify_all();
        return n;
    }
    size_t len() const {
        std::lock_guard<std::mutex> lk(s_->m);
        return s_->q.size();
    }

private:
    void attach() {
        if (!s_) return;
        std::lock_guard<std::mutex> lk(s_->m);
        ++s_->receivers;
    }
    void detach() {
        if (!s_) return;
        bool last;
        {
            std::lock_guard<std::mutex> lk(s_->m);
            last = --s_->receivers == 0;
        }
        if (last) s_->not_full.notify_all();
        s_.reset();
    }
    std::shared_ptr<Shared<T>> s_;
};

template <class T>
std::pair<Sender<T>, Receiver<T>> unbounded() {
    auto s = std::make_shared<Shared<T>>();
    return {Sender<T>(s), Receiver<T>(s)};
}
template <class T>
std::pair<Sender<T>, Receiver<T>> bounded(size_t cap) {
    auto s = std::make_shared<Shared<T>>();
    s->capacity = cap ? cap : 1;
    return {Sender<T>(s), Receiver<T>(s)};
}

}  // namespace channel
}  // namespace comms
