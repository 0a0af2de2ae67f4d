// node.hpp -- the reference's graph runtime surface, in C++.
//
// Mirrors (reference file:line):
//   trait Node: Send { start, call, is_connected }      src/node/mod.rs:94-98
//   enum NodeError { DataError, PermanentError, DataEnd, CommError }   :68-73
//   NodeReceiver<T> = Option<Receiver<T>>                src/prelude.rs:9
//   NodeSender<T>   = Vec<(Sender<T>, Option<T>)>        src/prelude.rs:10
//   #[derive(Node)] (+ #[aggregate]; #[pass_by_ref] is how C++ passes anyway)
//                                                        node_derive/src/lib.rs:53-222
//   connect_nodes! / connect_nodes_feedback!             src/node/mod.rs:150-156, :213-219
//   start_nodes! / start_nodes_threadpool!               src/node/mod.rs:276-284, :342-350
//   struct Graph                                         src/node/graph.rs:13-73
//
// A node is a struct with NodeReceiver / NodeSender fields and a
//     Result<Out> run(const In0&, const In1&, ...)
// member; `DeriveNode<Self, Out, In...>` supplies start()/call()/is_connected()
// exactly as the derive macro generates them: receive every input in
// declaration order (blocking), run, clone the result to every sender.
#pragma once

#include <functional>
#include <memory>
#include <optional>
#include <string>
#include <thread>
#include <tuple>
#include <utility>
#include <variant>
#include <vector>

#include "channel.hpp"

namespace comms {

enum class NodeError { DataError, PermanentError, DataEnd, CommError };

inline const char* to_string(NodeError e) {  // Display impl, src/node/mod.rs:75-85
    switch (e) {
        case NodeError::DataError: return "Node error: unable to access data";
        case NodeError::PermanentError: return "Node error: unable to continue executing node";
        case NodeError::DataEnd: return "Node error: end of data source";
        default: return "Node error: unable to establish comm channel";
    }
}

// Result<T, NodeError>
template <class T>
class Result {
public:
    Result(T v) : v_(std::move(v)) {}  // NOLINT: Ok(v)
    Result(NodeError e) : v_(e) {}     // NOLINT: Err(e)
    bool is_ok() const { return v_.index() == 0; }
    bool is_err() const { return !is_ok(); }
    T& value() { return std::get<0>(v_); }
    const T& value() const { return std::get<0>(v_); }
    NodeError error() const { return std::get<1>(v_); }

private:
    std::variant<T, NodeError> v_;
};
struct Unit {};
using Status = Result<Unit>;
inline Status Ok() { return Status(Unit{}); }

template <class T>
using NodeReceiver = std::optional<channel::Receiver<T>>;
template <class T>
using NodeSender = std::vector<std::pair<channel::Sender<T>, std::optional<T>>>;

struct Node {
    virtual ~Node() = default;
    virtual void start() = 0;
    virtual Status call() = 0;
    virtual bool is_connected() const = 0;
};

// connect_nodes!(n1, send, n2, recv)
template <class T>
void connect_nodes(NodeSender<T>& send, NodeReceiver<T>& recv) {
    auto ch = channel::unbounded<T>();
    send.emplace_back(std::move(ch.first), std::nullopt);
    recv = std::move(ch.second);
}
// connect_nodes_feedback!(n1, send, n2, recv, default): the default is sent once at start()
template <class T>
void connect_nodes_feedback(NodeSender<T>& send, NodeReceiver<T>& recv, T dflt) {
    auto ch = channel::unbounded<T>();
    send.emplace_back(std::move(ch.first), std::move(dflt));
    recv = std::move(ch.second);
}

namespace detail {
template <class T>
struct is_optional : std::false_type {};
template <class T>
struct is_optional<std::optional<T>> : std::true_type {};
// a node that can process a drained run of messages in one go declares
//   Result<std::vector<Out>> run_block(const std::vector<In>& ins)
template <class D, class = void>
struct has_run_block : std::false_type {};
template <class D>
struct has_run_block<D, std::void_t<decltype(&D::run_block)>> : std::true_type {};
}  // namespace detail

// What #[derive(Node)] generates.  `Derived` lists its fields through
//   auto receivers() { return std::tie(input0, input1, ...); }   (declaration order)
//   auto senders()   { return std::tie(output, ...); }
// and has run(const In&...) -> Result<Out>        (plain node), or
//         run(const In&...) -> Result<std::optional<Out>>   (#[aggregate]).
template <class Derived>
struct DeriveNode : Node {
    void start() override {
        auto& self = static_cast<Derived&>(*this);
        // feedback defaults go out once (node_derive/src/lib.rs:184-189)
        std::apply(
            [](auto&... snd) {
                (..., [&] {
                    for (auto& sv : snd)
                        if (sv.second) (void)sv.first.send(*sv.second);
                }());
            },
            self.senders());
        while (call().is_ok()) {
        }
    }

    // Per-sample nodes whose run() costs a device launch (FirNode, MixerNode, PulseNode,
    // FFTSampleNode) would turn every message into a launch + synchronisation.  When the node
    // offers run_block(), call() drains what is ALREADY queued behind the first message (recv, then
    // try_recv: it never waits for more), runs the block in one launch and sends the outputs one by
    // one in order -- every receiver sees exactly the message sequence of the reference's loop
    // (node_derive/src/lib.rs:200-211), a lone message is processed at once, and a busy producer
    // is followed at the rate of the kernel, not of the launch.
    static constexpr size_t kMaxBlock = size_t(1) << 16;

    Status call() override {
        auto& self = static_cast<Derived&>(*this);
        auto recvs = self.receivers();
        if constexpr (detail::has_run_block<Derived>::value && std::tuple_size_v<decltype(recvs)> == 1) {
            auto& r = std::get<0>(recvs);
            if (!r) return Status(NodeError::PermanentError);
            using In = typename std::remove_reference_t<decltype(r)>::value_type::value_type;
            static thread_local std::vector<In> block_in;  // a node lives on one thread; reused across calls
            block_in.clear();
            if (!r->recv_many(block_in, kMaxBlock)) return Status(NodeError::DataEnd);
            auto res = self.run_block(block_in);
            if (res.is_err()) return Status(res.error());
            bool ok = true;
            std::apply([&](auto&... snd) { (..., send_block(snd, res.value(), ok)); }, self.senders());
            return ok ? Ok() : Status(NodeError::CommError);
        } else {
            return recv_then_run(self, recvs, std::make_index_sequence<std::tuple_size_v<decltype(recvs)>>{});
        }
    }

    bool is_connected() const override {
        auto& self = const_cast<Derived&>(static_cast<const Derived&>(*this));
        bool ok = true;
        std::apply([&](auto&... r) { (..., (ok = ok && r.has_value())); }, self.receivers());
        std::apply([&](auto&... s) { (..., (ok = ok && !s.empty())); }, self.senders());
        return ok;
    }

private:
    template <class S, class V>
    static void send_block(S& snd, const std::vector<V>& outs, bool& ok) {
        for (auto& sv : snd) {
            if (!ok) return;
            std::vector<V> copy(outs);  // send(res.clone()) per message
            if (!sv.first.send_many(std::move(copy))) ok = false;
        }
    }

    template <class Tuple, size_t... I>
    Status recv_then_run(Derived& self, Tuple& recvs, std::index_sequence<I...>) {
        // Some(ref r) => r.recv().or(Err(DataEnd))?, None => return Err(PermanentError)
        std::tuple<std::optional<
            typename std::remove_reference_t<std::tuple_element_t<I, Tuple>>::value_type::value_type>...>
            in;
        NodeError err = NodeError::DataEnd;
        bool ok = true;
        (..., [&] {
            if (!ok) return;
            auto& r = std::get<I>(recvs);
            if (!r) {
                ok = false;
                err = NodeError::PermanentError;
                return;
            }
            auto v = r->recv();
            if (!v) {
                ok = false;
                err = NodeError::DataEnd;
                return;
            }
            std::get<I>(in) = std::move(v);
        }());
        if (!ok) return Status(err);
        auto res = self.run(*std::get<I>(in)...);
        if (res.is_err()) return Status(res.error());
        return send_all(self, std::move(res.value()));
    }

    template <class R>
    Status send_all(Derived& self, R&& res) {
        using RT = std::decay_t<R>;
        bool ok = true;
        if constexpr (std::is_same_v<RT, Unit>) {
            (void)self;
            (void)res;
        } else if constexpr (detail::is_optional<RT>::value) {  // #[aggregate]
            if (res) std::apply([&](auto&... snd) { (..., send_one(snd, *res, ok)); }, self.senders());
        } else {
            std::apply([&](auto&... snd) { (..., send_one(snd, res, ok)); }, self.senders());
        }
        return ok ? Ok() : Status(NodeError::CommError);
    }
    template <class S, class V>
    static void send_one(S& snd, const V& v, bool& ok) {
        for (auto& sv : snd)
            if (ok && !sv.first.send(v)) ok = false;  // send(res.clone())
    }
};

// start_nodes!(a, b, ...): one detached OS thread per node, which takes ownership
template <class... N>
void start_nodes(N&&... nodes) {
    (..., std::thread([n = std::make_shared<std::decay_t<N>>(std::move(nodes))]() { n->start(); }).detach());
}

// start_nodes_threadpool!: rayon::spawn.  Nodes block on their channels, so every
// node gets a pool thread of its own until the pool (hardware_concurrency workers)
// is exhausted; the remainder queue exactly as they would in rayon.
class ThreadPool {
public:
    static ThreadPool& global();
    void spawn(std::function<void()> f);
    ~ThreadPool();

private:
    explicit ThreadPool(unsigned n);
    void worker();
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    std::vector<std::thread> th_;
    bool stop_ = false;
};
inline ThreadPool::ThreadPool(unsigned n) {
    for (unsigned i = 0; i < (n ? n : 4u); ++i) th_.emplace_back([this] { worker(); });
}
inline ThreadPool& ThreadPool::global() {
    static ThreadPool* p = new ThreadPool(std::thread::hardware_concurrency());  // leaked: workers may outlive main
    return *p;
}
inline void ThreadPool::spawn(std::function<void()> f) {
    {
        std::lock_guard<std::mutex> lk(m_);
        q_.push_back(std::move(f));
    }
    cv_.notify_one();
}
inline void ThreadPool::worker() {
    for (;;) {
        std::function<void()> f;
        {
            std::unique_lock<std::mutex> lk(m_);
            cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
            if (stop_ && q_.empty()) return;
            f = std::move(q_.front());
            q_.pop_front();
        }
        f();
    }
}
inline ThreadPool::~ThreadPool() {
    {
        std::lock_guard<std::mutex> lk(m_);
        stop_ = true;
    }
    cv_.notify_all();
    for (auto& t : th_) t.join();
}
template <class... N>
void start_nodes_threadpool(N&&... nodes) {
    (..., ThreadPool::global().spawn([n = std::make_shared<std::decay_t<N>>(std::move(nodes))]() { n->start(); }));
}

// struct Graph (src/node/graph.rs): holds nodes + thread handles; optional bounded channels
class Graph {
public:
    explicit Graph(std::optional<size_t> channel_size = std::nullopt) : channel_size_(channel_size) {}
    void add_node(std::shared_ptr<Node> n) { nodes_.push_back(std::move(n)); }
    void add_nodes(std::vector<std::shared_ptr<Node>> ns) {
        for (auto& n : ns) add_node(std::move(n));
    }
    template <class T>
    void connect_nodes(NodeSender<T>& sender, NodeReceiver<T>& receiver, std::optional<T> dflt = std::nullopt) const {
        auto ch = channel_size_ ? channel::bounded<T>(*channel_size_) : channel::unbounded<T>();
        sender.emplace_back(std::move(ch.first), std::move(dflt));
        receiver = std::move(ch.second);
    }
    bool is_connected() const {
        for (auto& n : nodes_)
            if (!n->is_connected()) return false;
        return true;
    }
    // one thread per node, each running node.start() (graph.rs:65-73)
    void run_graph() {
        for (auto& n : nodes_) handles_.emplace_back([n] { n->start(); });
    }
    // not in the reference (its threads are never joined): lets tests end cleanly
    void join() {
        for (auto& h : handles_)
            if (h.joinable()) h.join();
        handles_.clear();
    }
    ~Graph() {
        for (auto& h : handles_)
            if (h.joinable()) h.detach();
    }

private:
    std::vector<std::shared_ptr<Node>> nodes_;
    std::vector<std::thread> handles_;
    std::optional<size_t> channel_size_;
};

}  // namespace comms
