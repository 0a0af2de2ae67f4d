// CPU-only tests of the C++ graph runtime (comms_rs_amd/host/comms/node.hpp),
// written after the reference's own runtime tests:
//   test_counter  (=55)          src/node/mod.rs:882-943
//   test_feedback (=512)         src/node/mod.rs:947-1010
//   test_fan_in                  src/node/mod.rs:770-876
//   test_aggregate_nodes         src/node/mod.rs:487-581
//   test_simple_graph (Graph)    src/node/mod.rs:420-480
//   tests/macro_tests.rs, tests/node_test.rs (derive + connect + start smoke)
// plus the error paths of the derive contract (node_derive/src/lib.rs:154-205).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <thread>

#include "../../comms_rs_amd/host/comms/node.hpp"

using namespace comms;

static int g_fail = 0;
#define CHECK(cond)                                                        \
    do {                                                                   \
        if (!(cond)) {                                                     \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_fail;                                                      \
        }                                                                  \
    } while (0)

// ---- counter
struct OneNode : DeriveNode<OneNode> {
    int count = 0;
    NodeSender<int> output;
    Result<int> run() { return ++count; }
    auto receivers() { return std::tie(); }
    auto senders() { return std::tie(output); }
};
struct CounterNode : DeriveNode<CounterNode> {
    NodeReceiver<int> input;
    int count = 0;
    NodeSender<int> output;
    Result<int> run(const int& v) { return count += v; }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};
static void test_counter() {
    OneNode one;
    CounterNode cnt;
    connect_nodes(one.output, cnt.input);
    std::thread t([&] {
        for (int i = 0; i < 10; ++i) CHECK(one.call().is_ok());
    });
    for (int i = 0; i < 10; ++i) CHECK(cnt.call().is_ok());
    t.join();
    CHECK(cnt.count == 55);
    CHECK(one.is_connected());
    CHECK(!cnt.is_connected());  // its output has no receiver attached
}

// ---- feedback
struct AddNode : DeriveNode<AddNode> {
    NodeReceiver<int> input;
    int count = 1;
    NodeSender<int> output;
    Result<int> run(const int& v) { return count += v; }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};
struct PrintNode : DeriveNode<PrintNode> {
    NodeReceiver<int> input;
    int count = 0;
    NodeSender<int> output;
    Result<int> run(const int& v) { return count = v; }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};
static void test_feedback() {
    AddNode add;
    PrintNode print;
    connect_nodes(add.output, print.input);
    connect_nodes_feedback(print.output, add.input, 0);
    // send print's feedback default by hand, as the reference test does, then step it
    for (auto& sv : print.output)
        if (sv.second) CHECK(sv.first.send(*sv.second));
    auto* pr = &print;
    start_nodes(std::move(add));
    for (int i = 0; i < 10; ++i) CHECK(pr->call().is_ok());
    CHECK(pr->count == 512);
}

// ---- fan-in + is_connected + PermanentError on an unconnected input
struct U32Source : DeriveNode<U32Source> {
    int left;
    NodeSender<unsigned> output;
    explicit U32Source(int n) : left(n) {}
    Result<unsigned> run() {
        if (left-- <= 0) return NodeError::DataEnd;
        return 1u;
    }
    auto receivers() { return std::tie(); }
    auto senders() { return std::tie(output); }
};
struct F64Source : DeriveNode<F64Source> {
    int left;
    NodeSender<double> output;
    explicit F64Source(int n) : left(n) {}
    Result<double> run() {
        if (left-- <= 0) return NodeError::DataEnd;
        return 2.0;
    }
    auto receivers() { return std::tie(); }
    auto senders() { return std::tie(output); }
};
struct DoubleInputNode : DeriveNode<DoubleInputNode> {
    NodeReceiver<unsigned> input1;
    NodeReceiver<double> input2;
    NodeSender<float> output;
    Result<float> run(const unsigned& x, const double& y) { return static_cast<float>(x + y); }
    auto receivers() { return std::tie(input1, input2); }
    auto senders() { return std::tie(output); }
};
struct CheckF32 : DeriveNode<CheckF32> {
    NodeReceiver<float> input;
    int seen = 0;
    Result<Unit> run(const float& x) {
        CHECK(x == 3.0f);
        ++seen;
        return Unit{};
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(); }
};
static void test_fan_in_and_data_end() {
    U32Source n1(100);
    F64Source n2(100);
    DoubleInputNode n3;
    CheckF32 n4;
    CHECK(!n3.is_connected());
    CHECK(n3.call().is_err() && n3.call().error() == NodeError::PermanentError);  // None receiver
    connect_nodes(n1.output, n3.input1);
    connect_nodes(n2.output, n3.input2);
    connect_nodes(n3.output, n4.input);
    CHECK(n3.is_connected() && n4.is_connected());
    start_nodes(std::move(n1), std::move(n2), std::move(n3));
    // sources end after 100 items -> their senders drop -> n3 sees DataEnd and exits ->
    // its sender drops -> n4 sees DataEnd: the failure cascade of SURVEY.md section 5
    Status st = Ok();
    while ((st = n4.call()).is_ok()) {
    }
    CHECK(st.error() == NodeError::DataEnd);
    CHECK(n4.seen == 100);
}

// ---- CommError when the downstream receiver has gone
static void test_comm_error() {
    OneNode one;
    {
        CounterNode cnt;
        connect_nodes(one.output, cnt.input);
        CHECK(one.call().is_ok());
    }  // cnt (and its Receiver) dropped
    Status st = one.call();
    CHECK(st.is_err() && st.error() == NodeError::CommError);
}

// ---- aggregate
using SharedVec = std::shared_ptr<std::vector<unsigned>>;  // Arc<Vec<u32>>
struct AggSource : DeriveNode<AggSource> {
    std::vector<unsigned> agg;
    int emitted = 0;
    NodeSender<SharedVec> output;
    Result<std::optional<SharedVec>> run() {
        if (emitted >= 50) return NodeError::DataEnd;
        if (agg.size() < 2) {
            agg.push_back(1);
            return std::optional<SharedVec>(std::nullopt);
        }
        auto v = std::make_shared<std::vector<unsigned>>(agg);
        agg.clear();
        ++emitted;
        return std::optional<SharedVec>(v);
    }
    auto receivers() { return std::tie(); }
    auto senders() { return std::tie(output); }
};
struct AddOne : DeriveNode<AddOne> {
    NodeReceiver<SharedVec> input;
    NodeSender<SharedVec> output;
    Result<SharedVec> run(const SharedVec& in) {
        auto y = std::make_shared<std::vector<unsigned>>(*in);  // Arc::make_mut on a shared Arc clones
        for (auto& z : *y) z += 1;
        return y;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};
struct CheckVec : DeriveNode<CheckVec> {
    NodeReceiver<SharedVec> input;
    int seen = 0;
    Result<Unit> run(const SharedVec& in) {
        CHECK((*in == std::vector<unsigned>{2, 2}));
        ++seen;
        return Unit{};
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(); }
};
static void test_aggregate() {
    AggSource n1;
    AddOne n2;
    CheckVec n3;
    connect_nodes(n1.output, n2.input);
    connect_nodes(n2.output, n3.input);
    start_nodes_threadpool(std::move(n1), std::move(n2));
    while (n3.call().is_ok()) {
    }
    CHECK(n3.seen == 50);
}

// ---- Graph with bounded channels (back-pressure) and a fan-out of one sender to two receivers
struct SinkSum : DeriveNode<SinkSum> {
    NodeReceiver<int> input;
    std::atomic<long>* total;
    int left;
    // Graph keeps its nodes (and their senders) alive after start() returns, exactly
    // like the reference's Arc<Mutex<dyn Node>> -- so no DataEnd cascade reaches a
    // sink; it stops itself after the expected number of items.
    SinkSum(std::atomic<long>* t, int expect) : total(t), left(expect) {}
    Result<Unit> run(const int& v) {
        total->fetch_add(v);
        if (--left == 0) return NodeError::DataEnd;
        return Unit{};
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(); }
};
struct CountTo : DeriveNode<CountTo> {
    int n, i = 0;
    NodeSender<int> output;
    explicit CountTo(int n_) : n(n_) {}
    Result<int> run() {
        if (i >= n) return NodeError::DataEnd;
        return ++i;
    }
    auto receivers() { return std::tie(); }
    auto senders() { return std::tie(output); }
};
static void test_graph_bounded_fanout() {
    std::atomic<long> a{0}, b{0};
    auto src = std::make_shared<CountTo>(1000);
    auto s1 = std::make_shared<SinkSum>(&a, 1000);
    auto s2 = std::make_shared<SinkSum>(&b, 1000);
    Graph g(2);  // Graph::new(Some(2)): bounded(2) channels
    CHECK(!src->is_connected());
    g.connect_nodes(src->output, s1->input);
    g.connect_nodes(src->output, s2->input);  // second sender on the same output: res.clone() to each
    g.add_nodes({src, s1, s2});
    CHECK(g.is_connected());
    g.run_graph();
    g.join();
    CHECK(a.load() == 500500 && b.load() == 500500);
}

// ---- channel unit behaviour
static void test_channel() {
    auto ch = channel::bounded<int>(1);
    CHECK(ch.first.send(1));
    std::atomic<bool> sent{false};
    std::thread t([&] {
        CHECK(ch.first.send(2));  // blocks until the receiver takes 1
        sent = true;
    });
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
    CHECK(!sent.load());
    CHECK(*ch.second.recv() == 1);
    t.join();
    CHECK(*ch.second.recv() == 2);
    CHECK(!ch.second.try_recv().has_value());
    {
        auto s2 = ch.first;  // clone
        channel::Sender<int> dead = std::move(ch.first);
        (void)dead;
        CHECK(s2.send(3));
    }  // both senders gone
    CHECK(*ch.second.recv() == 3);     // queued data still drains
    CHECK(!ch.second.recv().has_value());  // then RecvError
}

// ---- drained blocks: a per-sample node that offers run_block() must be indistinguishable, message
// for message, from the same node driven one sample at a time (node_derive/src/lib.rs:200-211)
struct RampSource : DeriveNode<RampSource> {
    int n, i = 0;
    NodeSender<int> output;
    explicit RampSource(int n_) : n(n_) {}
    Result<int> run() {
        if (i >= n) return NodeError::DataEnd;
        return i++;
    }
    auto receivers() { return std::tie(); }
    auto senders() { return std::tie(output); }
};
struct AccumulateSample : DeriveNode<AccumulateSample> {  // stateful: y[i] = y[i-1] + 3 x[i]
    NodeReceiver<int> input;
    NodeSender<long> output, tap;                    // two senders: every one gets every message
    long acc = 0;
    size_t calls = 0;
    Result<long> run(const int& v) {
        ++calls;
        return acc += 3L * v;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output, tap); }
};
struct AccumulateBlock : DeriveNode<AccumulateBlock> {    // the same node, offering run_block()
    NodeReceiver<int> input;
    NodeSender<long> output, tap;
    long acc = 0;
    size_t calls = 0, largest = 0;
    Result<long> run(const int& v) { return acc += 3L * v; }
    Result<std::vector<long>> run_block(const std::vector<int>& vs) {
        ++calls;
        if (vs.size() > largest) largest = vs.size();
        std::vector<long> out;
        for (int v : vs) out.push_back(acc += 3L * v);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output, tap); }
};
struct PairsBlock : DeriveNode<PairsBlock> {  // #[aggregate] in block form: one Vec per two inputs
    NodeReceiver<long> input;
    NodeSender<std::vector<long>> output;
    std::vector<long> held;
    Result<std::optional<std::vector<long>>> run(const long& v) {
        held.push_back(v);
        if (held.size() < 2) return std::optional<std::vector<long>>(std::nullopt);
        auto o = std::move(held);
        held.clear();
        return std::optional<std::vector<long>>(std::move(o));
    }
    Result<std::vector<std::vector<long>>> run_block(const std::vector<long>& vs) {
        std::vector<std::vector<long>> outs;
        for (long v : vs) {
            held.push_back(v);
            if (held.size() == 2) {
                outs.push_back(held);
                held.clear();
            }
        }
        return outs;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};
static void test_drained_blocks_equal_per_sample() {
    const int n = 200001;
    std::vector<long> per_sample, blocked, tapped;
    std::vector<std::vector<long>> pairs;
    {
        RampSource src(n);
        AccumulateSample acc;
        NodeReceiver<long> out, tap;
        connect_nodes(src.output, acc.input);
        connect_nodes(acc.output, out);
        connect_nodes(acc.tap, tap);
        std::thread a([&] { src.start(); src.output.clear(); });
        std::thread b([&] { acc.start(); acc.output.clear(); acc.tap.clear(); });
        while (auto v = out->recv()) per_sample.push_back(*v);
        a.join();
        b.join();
        CHECK(acc.calls == static_cast<size_t>(n));
    }
    {
        RampSource src(n);
        AccumulateBlock acc;
        PairsBlock pr;
        NodeReceiver<long> tap;
        NodeReceiver<std::vector<long>> out;
        connect_nodes(src.output, acc.input);
        connect_nodes(acc.output, pr.input);
        connect_nodes(acc.tap, tap);
        connect_nodes(pr.output, out);
        std::thread a([&] { src.start(); src.output.clear(); });
        std::thread b([&] { acc.start(); acc.output.clear(); acc.tap.clear(); });
        std::thread c([&] { pr.start(); pr.output.clear(); });
        std::thread d([&] { while (auto v = tap->recv()) tapped.push_back(*v); });
        while (auto v = out->recv()) pairs.push_back(*v);
        a.join();
        b.join();
        c.join();
        d.join();
        CHECK(acc.calls >= 1 && acc.calls <= static_cast<size_t>(n));
        std::printf("drained blocks: %zu calls for %d messages, largest block %zu\n", acc.calls, n, acc.largest);
    }
    for (auto& p : pairs) blocked.insert(blocked.end(), p.begin(), p.end());
    CHECK(pairs.size() == static_cast<size_t>(n / 2));          // the odd last sample stays held, as per sample
    CHECK(tapped == per_sample);                                 // every sender saw every message, in order
    per_sample.resize(blocked.size());
    CHECK(blocked == per_sample);
    // a lone message is processed at once: call() never waits for a second one
    {
        AccumulateBlock acc;
        NodeSender<int> in;
        NodeReceiver<long> out, tap;
        connect_nodes(in, acc.input);
        connect_nodes(acc.output, out);
        connect_nodes(acc.tap, tap);
        CHECK(in[0].first.send(5));
        CHECK(acc.call().is_ok());
        CHECK(*out->try_recv() == 15 && *tap->try_recv() == 15);
        in.clear();
        CHECK(acc.call().is_err() && acc.call().error() == NodeError::DataEnd);
    }
    // bounded channels: send_many respects the capacity (blocks instead of overfilling)
    {
        auto ch = channel::bounded<int>(4);
        std::thread t([&] { CHECK(ch.first.send_many(std::vector<int>{1, 2, 3, 4, 5, 6, 7, 8, 9})); });
        std::vector<int> got;
        while (got.size() < 9) {
            CHECK(ch.second.len() <= 4);
            CHECK(ch.second.recv_many(got, 3) > 0);
        }
        t.join();
        CHECK((got == std::vector<int>{1, 2, 3, 4, 5, 6, 7, 8, 9}));
    }
}

int main() {
    test_channel();
    test_drained_blocks_equal_per_sample();
    test_counter();
    test_feedback();
    test_fan_in_and_data_end();
    test_comm_error();
    test_aggregate();
    test_graph_bounded_fanout();
    if (g_fail) {
        std::fprintf(stderr, "%d check(s) failed\n", g_fail);
        return 1;
    }
    std::puts("host graph tests: all passed");
    return 0;
}
