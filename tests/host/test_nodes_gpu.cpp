// GPU tests of the C++ DSP nodes in a graph, written like the reference's node
// tests: source -> node-under-test -> CheckNode, goldens inline
//   FIR sample / batch      src/filter/fir_node.rs:235-338, :342-449
//   FFT batch / sample      src/fft/fft_node.rs:180-263, :266-345
//   mixer (phase 0 / 0.1)   src/mixer.rs:160-246, :250-336
//   pulse (rect x4)         src/pulse.rs:105-209
//   decimate / upsample     src/util/resample_node.rs:139-175
//   timing estimator / NCO  src/demodulation/timing_estimator.rs:178-229, nco.rs:71-77
// plus a device-resident FIR -> mixer -> decimate graph (DeviceBuf messages).
// Needs an MI355X (libcomms_hip has no CPU fallback).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <thread>

#include "../../comms_rs_amd/host/comms/nodes.hpp"

using namespace comms;
using C = Complex32;

static int g_fail = 0;
#define CHECK(cond)                                                                       \
    do {                                                                                  \
        if (!(cond)) {                                                                    \
            std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); \
            ++g_fail;                                                                     \
        }                                                                                 \
    } while (0)

// replays a literal Vec, then ends (the reference's sources emit zeros forever)
template <class T>
struct Replay : DeriveNode<Replay<T>> {
    std::vector<T> items;
    size_t i = 0;
    NodeSender<T> output;
    explicit Replay(std::vector<T> v) : items(std::move(v)) {}
    Result<T> run() {
        if (i >= items.size()) return NodeError::DataEnd;
        return items[i++];
    }
    auto receivers() { return std::tie(); }
    auto senders() { return std::tie(output); }
};
template <class T>
struct Collect : DeriveNode<Collect<T>> {
    NodeReceiver<T> input;
    std::vector<T> got;
    Result<Unit> run(const T& v) {
        got.push_back(v);
        return Unit{};
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(); }
};
template <class Src, class Dut, class Chk>
static void pump(Src src, Dut dut, Chk& chk) {
    connect_nodes(src.output, dut.input);
    connect_nodes(dut.output, chk.input);
    start_nodes(std::move(src), std::move(dut));
    while (chk.call().is_ok()) {
    }
}
static bool close(C a, C b, float tol) { return std::abs(a - b) < tol; }

static const std::vector<C> kFirIn = {{1, 2}, {3, 4}, {5, 6}, {7, 8}, {9, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
static const std::vector<C> kFirTaps = {{9, 0}, {8, 7}, {6, 5}, {4, 3}, {2, 1}};
static const std::vector<C> kFirOut = {{9, 18},   {21, 59},  {37, 124}, {57, 205}, {81, 204},
                                       {78, 196}, {62, 115}, {42, 50},  {18, 9}};

static void test_fir_nodes() {
    Collect<C> chk;
    pump(Replay<C>(kFirIn), FirNode(kFirTaps), chk);
    CHECK(chk.got.size() == 10);
    for (size_t i = 0; i < kFirOut.size() && i < chk.got.size(); ++i) CHECK(chk.got[i] == kFirOut[i]);

    // batch node, two samples per message (the reference's batch test feeds 2 at a time)
    std::vector<std::vector<C>> batches;
    for (size_t i = 0; i < kFirIn.size(); i += 2) batches.push_back({kFirIn[i], kFirIn[i + 1]});
    Collect<std::vector<C>> chk2;
    pump(Replay<std::vector<C>>(batches), BatchFirNode(kFirTaps), chk2);
    std::vector<C> flat;
    for (auto& b : chk2.got) flat.insert(flat.end(), b.begin(), b.end());
    CHECK(flat.size() == 10);
    for (size_t i = 0; i < kFirOut.size() && i < flat.size(); ++i) CHECK(flat[i] == kFirOut[i]);
}

static void test_integer_nodes() {
    // the reference's own FIR and pulse tests, on the type they use: Complex<i16> (fir_node.rs:235-338, pulse.rs:105-209)
    using I = Complex16;
    std::vector<I> in, taps, want;
    for (auto& v : kFirIn) in.emplace_back(static_cast<int16_t>(v.real()), static_cast<int16_t>(v.imag()));
    for (auto& v : kFirTaps) taps.emplace_back(static_cast<int16_t>(v.real()), static_cast<int16_t>(v.imag()));
    for (auto& v : kFirOut) want.emplace_back(static_cast<int16_t>(v.real()), static_cast<int16_t>(v.imag()));
    Collect<I> chk;
    pump(Replay<I>(in), FirNodeI16(taps), chk);
    CHECK(chk.got.size() == 10);
    for (size_t i = 0; i < want.size() && i < chk.got.size(); ++i) CHECK(chk.got[i] == want[i]);
    const std::vector<I> sym = {{-1, -1}, {1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    Collect<std::vector<I>> chk2;
    pump(Replay<I>(sym), PulseNodeI16(std::vector<I>(4, I(1, 0)), 4), chk2);
    CHECK(chk2.got.size() == 5);
    for (size_t i = 0; i < chk2.got.size(); ++i) {
        CHECK(chk2.got[i].size() == 4);
        for (auto& v : chk2.got[i]) CHECK(v == sym[i]);   // rect taps x4: every symbol held four samples (pulse.rs:162-183)
    }
    // wrap-around: 300 * 300 = 90000 = 24464 (mod 2^16), as i16 arithmetic gives in a release build of the reference
    BatchFirNodeI16 w({I(300, 0)});
    auto r = w.run({I(300, 0)});
    CHECK(r.is_ok() && r.value()[0] == I(static_cast<int16_t>(90000 & 0xffff), 0));
}

static void test_f64_nodes() {
    // the same two reference tests on Complex<f64> (the type of the reference's batch_fir doc example, fir.rs:68-86): the
    // goldens are small integers, which f64 holds exactly, and the f64 kernel does the reference's arithmetic operation for
    // operation -- equality, not a tolerance
    using D = Complex64;
    std::vector<D> in, taps, want;
    for (auto& v : kFirIn) in.emplace_back(v.real(), v.imag());
    for (auto& v : kFirTaps) taps.emplace_back(v.real(), v.imag());
    for (auto& v : kFirOut) want.emplace_back(v.real(), v.imag());
    Collect<D> chk;
    pump(Replay<D>(in), FirNodeF64(taps), chk);
    CHECK(chk.got.size() == 10);
    for (size_t i = 0; i < want.size() && i < chk.got.size(); ++i) CHECK(chk.got[i] == want[i]);
    Collect<std::vector<D>> chkb;
    pump(Replay<std::vector<D>>({std::vector<D>(in.begin(), in.begin() + 5), std::vector<D>(in.begin() + 5, in.end())}),
         BatchFirNodeF64(taps), chkb);
    std::vector<D> flat;
    for (auto& b : chkb.got) flat.insert(flat.end(), b.begin(), b.end());
    CHECK(flat.size() == 10);
    for (size_t i = 0; i < want.size() && i < flat.size(); ++i) CHECK(flat[i] == want[i]);
    const std::vector<D> sym = {{-1, -1}, {1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    Collect<std::vector<D>> chk2;
    pump(Replay<D>(sym), PulseNodeF64(std::vector<D>(4, D(1, 0)), 4), chk2);
    CHECK(chk2.got.size() == 5);
    for (size_t i = 0; i < chk2.got.size(); ++i) {
        CHECK(chk2.got[i].size() == 4);
        for (auto& v : chk2.got[i]) CHECK(v == sym[i]);
    }
}

static void test_fft_nodes() {
    std::vector<C> in, want = {{5.5f, 5.5f},           {-2.03884f, 1.03884f}, {-1.18819f, 0.18819f},
                               {-0.86327f, -0.13673f}, {-0.66246f, -0.33754f}, {-0.5f, -0.5f},
                               {-0.33754f, -0.66246f}, {-0.13673f, -0.86327f}, {0.18819f, -1.18819f},
                               {1.03884f, -2.03884f}};
    for (int i = 1; i <= 10; ++i) in.push_back(C(0.1f * i, 0.1f * i));
    Collect<std::vector<C>> chk;
    pump(Replay<std::vector<C>>({in, in}), FFTBatchNode(10, false), chk);
    CHECK(chk.got.size() == 2);
    for (auto& v : chk.got)
        for (size_t i = 0; i < 10 && v.size() == 10; ++i) CHECK(close(v[i], want[i], 1e-5f));
    // sample node (#[aggregate]): 25 pushes -> exactly two transforms out
    std::vector<C> stream;
    for (int r = 0; r < 25; ++r) stream.push_back(in[r % 10]);
    Collect<std::vector<C>> chk2;
    pump(Replay<C>(stream), FFTSampleNode(10, false), chk2);
    CHECK(chk2.got.size() == 2);
    for (auto& v : chk2.got)
        for (size_t i = 0; i < 10 && v.size() == 10; ++i) CHECK(close(v[i], want[i], 1e-5f));
    // wrong length -> DataError, node stops (the reference panics there)
    FFTBatchNode bad(16, false);
    auto r = bad.run(std::vector<C>(15));
    CHECK(r.is_err() && r.error() == NodeError::DataError);
}

static void test_mixer_nodes() {
    const std::vector<C> in = {{1, 2}, {3, 4}, {5, 6}, {7, 8}, {9, 0}};
    const std::vector<C> want0 = {{1.0f, 2.0f},
                                  {2.486574736f, 4.337850399f},
                                  {3.388313374f, 7.036997405f},
                                  {3.643356072f, 9.986288426f},
                                  {7.932508585f, 4.251506503f}};
    const std::vector<C> want1 = {{0.795337332f, 2.089841747f},
                                  {2.041089794f, 4.564422467f},
                                  {2.668858427f, 7.340108630f},
                                  {2.628189174f, 10.300127265f},
                                  {7.468436663f, 5.022196114f}};
    Collect<C> a, b;
    pump(Replay<C>(in), MixerNode(0.123), a);
    pump(Replay<C>(in), MixerNode(0.123, 0.1), b);
    CHECK(a.got.size() == 5 && b.got.size() == 5);
    for (size_t i = 0; i < 5 && a.got.size() == 5 && b.got.size() == 5; ++i) {
        CHECK(close(a.got[i], want0[i], 2e-6f));  // f64 goldens at 1e-6; f32 rounding at |y|~10 is 5e-7
        CHECK(close(b.got[i], want1[i], 2e-6f));
    }
    // MixerNode<f64>: the reference's tests themselves (mixer.rs:160-246, :250-336) at their own 1e-6
    const std::vector<Complex64> in64 = {{1, 2}, {3, 4}, {5, 6}, {7, 8}, {9, 0}};
    const std::vector<Complex64> w0 = {{1.0, 2.0}, {2.486574736, 4.337850399}, {3.388313374, 7.036997405},
                                       {3.643356072, 9.986288426}, {7.932508585, 4.251506503}};
    const std::vector<Complex64> w1 = {{0.795337332, 2.089841747}, {2.041089794, 4.564422467}, {2.668858427, 7.340108630},
                                       {2.628189174, 10.300127265}, {7.468436663, 5.022196114}};
    Collect<Complex64> a64, b64;
    pump(Replay<Complex64>(in64), MixerNode64(0.123), a64);
    pump(Replay<Complex64>(in64), MixerNode64(0.123, 0.1), b64);
    CHECK(a64.got.size() == 5 && b64.got.size() == 5);
    for (size_t i = 0; i < 5 && a64.got.size() == 5 && b64.got.size() == 5; ++i) {
        CHECK(std::abs(a64.got[i].real() - w0[i].real()) < 1e-6 && std::abs(a64.got[i].imag() - w0[i].imag()) < 1e-6);
        CHECK(std::abs(b64.got[i].real() - w1[i].real()) < 1e-6 && std::abs(b64.got[i].imag() - w1[i].imag()) < 1e-6);
    }
}

static void test_pulse_node() {
    const std::vector<C> sym = {{-1, -1}, {1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    Collect<std::vector<C>> chk;
    pump(Replay<C>(sym), PulseNode(rect_taps(4), 4), chk);
    CHECK(chk.got.size() == 5);
    for (size_t s = 0; s < chk.got.size(); ++s) {
        CHECK(chk.got[s].size() == 4);
        for (auto& v : chk.got[s]) CHECK(v == sym[s]);
    }
}

static void test_resample_nodes() {
    std::vector<int> v1 = {1, 2, 3, 4, 5, 6}, out;
    CHECK(DecimateNode<int>(2).decimate(v1, out) == COMMS_OK && out == std::vector<int>({1, 3, 5}));
    CHECK(DecimateNode<int>(100).decimate(v1, out) == COMMS_OK && out == std::vector<int>({1}));
    CHECK(DecimateNode<int>(0).decimate(v1, out) == COMMS_OK && out == v1);
    CHECK(DecimateNode<int>(1).decimate(v1, out) == COMMS_OK && out == v1);
    std::vector<int> v2 = {1, 2, 3, 4};
    CHECK(UpsampleNode<int>(4).upsample(v2, out) == COMMS_OK &&
          out == std::vector<int>({1, 0, 0, 0, 2, 0, 0, 0, 3, 0, 0, 0, 4, 0, 0, 0}));
    CHECK(UpsampleNode<int>(0).upsample(v2, out) == COMMS_OK && out == v2);
    CHECK(UpsampleNode<int>(1).upsample(v2, out) == COMMS_OK && out == v2);
    Collect<std::vector<float>> chk;
    pump(Replay<std::vector<float>>({{1, 2, 3, 4, 5, 6, 7, 8}}), DecimateNode<float>(3), chk);
    CHECK(chk.got.size() == 1 && chk.got[0] == std::vector<float>({1, 4, 7}));
}

static void test_fm_node() {
    // no reference test exists; pin the signed-zero first-sample edge and a tone
    Collect<std::vector<float>> chk;
    std::vector<C> tone(4096);
    for (size_t i = 0; i < tone.size(); ++i) tone[i] = std::polar(1.0f, 0.25f * static_cast<float>(i));
    pump(Replay<std::vector<C>>({{C(-1, -1)}, tone}), FMDemodNode(), chk);
    CHECK(chk.got.size() == 2);
    if (chk.got.size() == 2) {
        CHECK(chk.got[0].size() == 1 && chk.got[0][0] == static_cast<float>(M_PI));
        for (size_t i = 1; i < chk.got[1].size(); ++i) CHECK(std::fabs(chk.got[1][i] - 0.25f) < 1e-3f);
    }
}

// The 63 taps examples/fm_radio.rs:30-52 ships, read from the fixture (tests/golden/reference_kats.json: data the
// reference holds, transcribed with its citation).
static std::vector<C> fm_radio_taps() {
    std::vector<C> taps;
    for (const char* path : {"tests/golden/reference_kats.json", "../../tests/golden/reference_kats.json"}) {
        std::FILE* f = std::fopen(path, "rb");
        if (!f) continue;
        std::string txt;
        char buf[4096];
        size_t got;
        while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) txt.append(buf, got);
        std::fclose(f);
        size_t p = txt.find("\"fm_radio_taps\"");
        if (p == std::string::npos) break;
        p = txt.find("\"taps_re\"", p);
        p = txt.find('[', p);
        const size_t end = txt.find(']', p);
        const char* q = txt.c_str() + p + 1;
        while (q < txt.c_str() + end) {
            char* e = nullptr;
            const double v = std::strtod(q, &e);
            if (e == q) break;
            taps.push_back(C(static_cast<float>(v), 0.f));
            q = e;
            while (q < txt.c_str() + end && (*q == ',' || *q == ' ' || *q == '\n')) ++q;
        }
        break;
    }
    return taps;
}

// examples/fm_radio.rs:144-164 as a graph, with the example's own filter: bytes -> ConvertNode -> BatchFirNode<f32>
// -> DecimateNode<Complex<f32>>(5) -> FMDemodNode -> Convert2Node -> BatchFirNode<f32> -> Convert3Node ->
// DecimateNode<f32>(5) -> sink.  The three Convert nodes are the example's own (user-defined) nodes.
struct ConvertU8 : DeriveNode<ConvertU8> {  // fm_radio.rs:62-91
    NodeReceiver<std::vector<uint8_t>> input;
    NodeSender<std::vector<C>> output;
    Result<std::vector<C>> run(const std::vector<uint8_t>& b) {
        std::vector<C> out(b.size() / 2);
        for (size_t i = 0; i < out.size(); ++i) out[i] = C((b[2 * i] - 127.5f) / 127.5f, (b[2 * i + 1] - 127.5f) / 127.5f);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};
struct Convert2 : DeriveNode<Convert2> {  // fm_radio.rs:93-117
    NodeReceiver<std::vector<float>> input;
    NodeSender<std::vector<C>> output;
    Result<std::vector<C>> run(const std::vector<float>& v) {
        std::vector<C> out(v.size());
        for (size_t i = 0; i < v.size(); ++i) out[i] = C(v[i], 0.f);
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};
struct Convert3 : DeriveNode<Convert3> {  // fm_radio.rs:119-141
    NodeReceiver<std::vector<C>> input;
    NodeSender<std::vector<float>> output;
    Result<std::vector<float>> run(const std::vector<C>& v) {
        std::vector<float> out(v.size());
        for (size_t i = 0; i < v.size(); ++i) out[i] = v[i].real();
        return out;
    }
    auto receivers() { return std::tie(input); }
    auto senders() { return std::tie(output); }
};

static void test_fm_radio_literal_graph() {
    const std::vector<C> taps = fm_radio_taps();
    CHECK(taps.size() == 63);
    if (taps.size() != 63) return;
    for (size_t k = 0; k < 63; ++k) CHECK(taps[k] == taps[62 - k]);  // the example's filter is symmetric
    // four radio blocks (RadioRxNode::new(rtlsdr, 0, 262144): 131072 samples each) of an FM tone: 1 kHz-like
    // modulation on a carrier offset, quantised to the radio's bytes
    const size_t blk = 131072, nblk = 4;
    std::vector<std::vector<uint8_t>> blocks(nblk, std::vector<uint8_t>(2 * blk));
    for (size_t b = 0; b < nblk; ++b)
        for (size_t i = 0; i < blk; ++i) {
            const double n = static_cast<double>(b * blk + i);
            const double ph = 2.0 * M_PI * 0.02 * n + 4.0 * std::cos(2.0 * M_PI * n / 5000.0);
            blocks[b][2 * i] = static_cast<uint8_t>(std::lround(127.5 + 100.0 * std::cos(ph)));
            blocks[b][2 * i + 1] = static_cast<uint8_t>(std::lround(127.5 + 100.0 * std::sin(ph)));
        }
    // the graph, wired as the example wires it (fm_radio.rs:156-164)
    Replay<std::vector<uint8_t>> sdr(blocks);
    ConvertU8 convert;
    BatchFirNode filt1(taps), filt2(taps);
    DecimateNode<C> dec1(5);
    FMDemodNode fm;
    Convert2 convert2;
    Convert3 convert3;
    DecimateNode<float> dec2(5);
    Collect<std::vector<float>> audio;
    connect_nodes(sdr.output, convert.input);
    connect_nodes(convert.output, filt1.input);
    connect_nodes(filt1.output, dec1.input);
    connect_nodes(dec1.output, fm.input);
    connect_nodes(fm.output, convert2.input);
    connect_nodes(convert2.output, filt2.input);
    connect_nodes(filt2.output, convert3.input);
    connect_nodes(convert3.output, dec2.input);
    connect_nodes(dec2.output, audio.input);
    start_nodes(std::move(sdr), std::move(convert), std::move(filt1), std::move(dec1), std::move(fm), std::move(convert2),
                std::move(filt2), std::move(convert3), std::move(dec2));
    while (audio.call().is_ok()) {
    }
    CHECK(audio.got.size() == nblk);
    // the same blocks through the same node types called directly, state carried across blocks
    BatchFirNode f1(taps), f2(taps);
    FMDemodNode fmd;
    ConvertU8 cv;
    Convert2 c2;
    Convert3 c3;
    double worst = 0.0;
    for (size_t b = 0; b < nblk && b < audio.got.size(); ++b) {
        const auto a = fmd.run(DecimateNode<C>(5).run(f1.run(cv.run(blocks[b]).value()).value()).value()).value();
        const auto want = DecimateNode<float>(5).run(c3.run(f2.run(c2.run(a).value()).value()).value()).value();
        CHECK(want.size() == audio.got[b].size());
        for (size_t i = 0; i < want.size() && i < audio.got[b].size(); ++i)
            worst = std::fmax(worst, std::fabs(static_cast<double>(want[i]) - audio.got[b][i]));
    }
    CHECK(worst == 0.0);  // the graph adds nothing: same nodes, same launches, same state
    // what comes out is the instantaneous frequency: 2 pi 0.02 * 5 per decimated sample, modulated by the tone, low-passed
    if (!audio.got.empty() && audio.got.back().size() > 1000) {
        double mean = 0.0;
        const auto& v = audio.got.back();
        for (size_t i = 200; i < v.size(); ++i) mean += v[i];
        mean /= static_cast<double>(v.size() - 200);
        CHECK(std::fabs(mean - 2.0 * M_PI * 0.02 * 5.0 * 1.0363602) < 0.02);  // x the filter's DC gain (sum of the taps)
    }
    // the front half as ONE launch reading the radio's bytes (comms_chain_*, u8 load stage) agrees with the three nodes
    comms_chain_t* front = nullptr;
    CHECK(comms_chain_create_ex(0.0, 0.0, c32(taps.data()), taps.size(), 5, COMMS_CHAIN_FM_DEMOD, 0, &front) == COMMS_OK);
    CHECK(comms_chain_set_input_format(front, COMMS_IQ_U8, 1.0f) == COMMS_OK);
    BatchFirNode g1(taps);
    FMDemodNode gfm;
    const size_t nw = blk / 5 * 5;  // the fused chain takes whole decimation periods
    std::vector<uint8_t> whole(blocks[0].begin(), blocks[0].begin() + 2 * nw);
    const auto ref = gfm.run(DecimateNode<C>(5).run(g1.run(cv.run(whole).value()).value()).value()).value();
    std::vector<float> fused(nw / 5);
    CHECK(comms_chain_run(front, reinterpret_cast<const comms_c32*>(whole.data()), nw, fused.data()) == COMMS_OK);
    double wf = 0.0;
    for (size_t i = 1; i < fused.size() && i < ref.size(); ++i) {
        double e = std::fabs(static_cast<double>(fused[i]) - ref[i]);
        if (e > M_PI) e = 2.0 * M_PI - e;
        wf = std::fmax(wf, e);
    }
    CHECK(fused.size() == ref.size() && wf < 2e-3);
    comms_chain_destroy(front);
}

static void test_demod_nodes() {
    // timing_estimator.rs:178-229: QPSK x10 zero-stuffed through a 101-tap RRC; the estimate of
    // samples[2..] must land within 0.01 samples of -2.  Symbols from an LCG (the reference
    // draws them from SmallRng).
    const uint32_t sps = 10;
    auto rrc = rrc_taps(sps * 10 + 1, static_cast<double>(sps), 0.5);
    std::vector<Complex64> sig(1000 * sps, Complex64(0, 0)), shaped(sig.size());
    uint32_t lcg = 12345;
    for (size_t k = 0; k < 1000; ++k) {
        lcg = lcg * 1664525u + 1013904223u;
        sig[k * sps] = std::polar(1.0, 2.0 * M_PI * ((lcg >> 24) & 3) / 4.0 + M_PI / 4.0);
    }
    for (size_t i = 0; i < sig.size(); ++i) {
        Complex64 acc(0, 0);
        for (size_t k = 0; k < rrc.size() && k <= i; ++k) acc += Complex64(rrc[k].real(), 0.0) * sig[i - k];
        shaped[i] = acc;
    }
    Collect<double> est;
    pump(Replay<std::vector<Complex64>>({std::vector<Complex64>(shaped.begin() + 2, shaped.end())}),
         TimingEstimatorNode(sps, 5, 0.5), est);
    CHECK(est.got.size() == 1);
    if (est.got.size() == 1) CHECK(std::fabs(2.0 + est.got[0]) < 0.01);
    bool threw = false;
    try {
        TimingEstimatorNode bad(10, 5, 1.5);
    } catch (const std::runtime_error&) {
        threw = true;
    }
    CHECK(threw);

    // NCO: two blocks, phase carried across (nco.rs:71-77)
    std::vector<double> e0(1000, 0.01), e1(500, -0.02);
    Collect<std::vector<Complex64>> osc;
    pump(Replay<std::vector<double>>({e0, e1}), BatchNcoNode(0.1, M_PI / 4.0), osc);
    CHECK(osc.got.size() == 2);
    if (osc.got.size() == 2) {
        double ph = M_PI / 4.0, worst = 0.0;
        for (size_t b = 0; b < 2; ++b)
            for (size_t i = 0; i < osc.got[b].size(); ++i) {
                ph += 0.1 + (b == 0 ? 0.01 : -0.02);
                if (ph > 2.0 * M_PI) ph -= 2.0 * M_PI;
                worst = std::fmax(worst, std::abs(osc.got[b][i] - std::polar(1.0, ph)));
            }
        CHECK(worst < 1e-10);
    }
    auto q = qfilt_taps(21, 0.25, 2);
    CHECK(q.size() == 21 && q[6] == 0.0625 && std::fabs(q[10] - 0.07957747154594767) < 1e-16);
}

static void test_device_resident_graph() {
    // FIR -> mixer -> decimate with DeviceBuf messages; compare with the host-vector nodes
    const size_t n = 1 << 16;
    std::vector<C> x(n);
    comms_synth_iq_host(c32(x.data()), n, 0, 0xC0FFEE);
    auto taps = rrc_taps(255, 8.0, 0.35);
    BatchFirNode f(taps);
    BatchMixerNode m(0.6283185307179586);
    DecimateNode<C> d(8);
    auto y = d.run(m.run(f.run(x).value()).value()).value();
    std::vector<C> h0(x.begin(), x.begin() + n / 2), h1(x.begin() + n / 2, x.end());
    Collect<DeviceBuf<C>> chk;
    Replay<DeviceBuf<C>> src({DeviceBuf<C>::from_host(h0), DeviceBuf<C>::from_host(h1)});
    BatchFirNodeDev fd(taps);
    BatchMixerNodeDev md(0.6283185307179586);
    DecimateNodeDev dd(8);
    connect_nodes(src.output, fd.input);
    connect_nodes(fd.output, md.input);
    connect_nodes(md.output, dd.input);
    connect_nodes(dd.output, chk.input);
    start_nodes(std::move(src), std::move(fd), std::move(md), std::move(dd));
    while (chk.call().is_ok()) {
    }
    CHECK(chk.got.size() == 2);
    std::vector<C> got;
    for (auto& b : chk.got) {
        auto v = b.to_host();
        got.insert(got.end(), v.begin(), v.end());
    }
    CHECK(got.size() == y.size());
    float worst = 0;
    for (size_t i = 0; i < got.size() && i < y.size(); ++i) worst = std::max(worst, std::abs(got[i] - y[i]));
    // one call vs two calls of the same kernels: rounding-level differences only
    CHECK(worst < 1e-5f * 15.0f);
}

static void test_device_resident_stream_many_messages() {
    // 160 messages of 2^17 samples through FIR -> mixer -> decimate, every node on its own stream and thread,
    // no device synchronisation anywhere: ordering rides on the buffers' events, the buffers come from the
    // cache (each message's memory is recycled while later messages are still in flight).  Equal to the
    // same stream through the host-vector nodes.
    const size_t m = 1 << 17, k = 160;
    auto taps = rrc_taps(63, 4.0, 0.25);
    std::vector<DeviceBuf<C>> msgs;
    std::vector<C> x(m * k);
    comms_synth_iq_host(c32(x.data()), x.size(), 0, 7);
    BatchFirNode f(taps);
    BatchMixerNode mx(0.2);
    DecimateNode<C> d(4);
    std::vector<C> want;
    for (size_t i = 0; i < k; ++i) {
        std::vector<C> part(x.begin() + i * m, x.begin() + (i + 1) * m);
        msgs.push_back(DeviceBuf<C>::from_host(part));
        auto y = d.run(mx.run(f.run(part).value()).value()).value();
        want.insert(want.end(), y.begin(), y.end());
    }
    Collect<DeviceBuf<C>> chk;
    Replay<DeviceBuf<C>> src(std::move(msgs));
    BatchFirNodeDev fd(taps);
    BatchMixerNodeDev md(0.2);
    DecimateNodeDev dd(4);
    connect_nodes(src.output, fd.input);
    connect_nodes(fd.output, md.input);
    connect_nodes(md.output, dd.input);
    connect_nodes(dd.output, chk.input);
    const auto t0 = std::chrono::steady_clock::now();
    start_nodes(std::move(src), std::move(fd), std::move(md), std::move(dd));
    while (chk.call().is_ok()) {
    }
    CHECK(chk.got.size() == k);
    std::vector<C> got;
    for (auto& b : chk.got) {
        auto v = b.to_host();   // waits for that message's producer only
        got.insert(got.end(), v.begin(), v.end());
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("device-resident graph FIR -> mixer -> decimate: %zu messages of %zu samples in %.3f s (incl. D2H of the results)\n", k, m, dt);
    CHECK(got.size() == want.size());
    size_t bad = 0;
    for (size_t i = 0; i < got.size() && i < want.size(); ++i) bad += got[i] != want[i];
    CHECK(bad == 0);   // same kernels, same batch cuts: bit-identical
    chk.got.clear();
    CHECK(comms_buf_pool_trim(0) == COMMS_OK);
}

static void test_device_resident_fan_out() {
    // One producer, two consumers of the SAME buffers (NodeSender clones once per sender, node_derive/src/lib.rs:156
    // = comms_buf_retain): FIR -> {mixer, decimate}.  Each consumer waits for the producer's event on its own stream
    // and records its own use; a buffer's memory is recycled only after both launches -- while 120 more messages are
    // in flight behind it.  Results equal the host-vector nodes bit for bit.
    const size_t m = 1 << 16, k = 120;
    auto taps = rrc_taps(63, 4.0, 0.25);
    std::vector<C> x(m * k);
    comms_synth_iq_host(c32(x.data()), x.size(), 0, 11);
    BatchFirNode f(taps);
    BatchMixerNode mx(0.3, 1.0);
    DecimateNode<C> d(5);
    std::vector<C> want_m, want_d;
    std::vector<DeviceBuf<C>> msgs;
    for (size_t i = 0; i < k; ++i) {
        std::vector<C> part(x.begin() + i * m, x.begin() + (i + 1) * m);
        msgs.push_back(DeviceBuf<C>::from_host(part));
        auto y = f.run(part).value();
        auto a = mx.run(y).value();
        auto b = d.run(y).value();
        want_m.insert(want_m.end(), a.begin(), a.end());
        want_d.insert(want_d.end(), b.begin(), b.end());
    }
    Replay<DeviceBuf<C>> src(std::move(msgs));
    BatchFirNodeDev fd(taps);
    BatchMixerNodeDev md(0.3, 1.0);
    DecimateNodeDev dd(5);
    Collect<DeviceBuf<C>> cm, cd;
    connect_nodes(src.output, fd.input);
    connect_nodes(fd.output, md.input);
    connect_nodes(fd.output, dd.input);   // second sender on the same output: every message is cloned
    connect_nodes(md.output, cm.input);
    connect_nodes(dd.output, cd.input);
    start_nodes(std::move(src), std::move(fd), std::move(md), std::move(dd));
    std::thread t([&] {
        while (cd.call().is_ok()) {
        }
    });
    while (cm.call().is_ok()) {
    }
    t.join();
    CHECK(cm.got.size() == k && cd.got.size() == k);
    std::vector<C> got_m, got_d;
    for (auto& b : cm.got) {
        auto v = b.to_host();
        got_m.insert(got_m.end(), v.begin(), v.end());
    }
    for (auto& b : cd.got) {
        auto v = b.to_host();
        got_d.insert(got_d.end(), v.begin(), v.end());
    }
    CHECK(got_m == want_m);
    CHECK(got_d == want_d);
}

// One stream cut over two chain nodes (as two GPUs would hold them): the second shard's node is created with the
// oscillator phase of the first PREFIX sample, runs the prefix (outputs dropped) and then its shard -- and the two
// shards' outputs are the single node's over the whole stream.  Also the node-by-node form: FIR state from the halo,
// mixer phase in closed form.  (The prefix / halo is what one rank receives from its left neighbour.)
static void test_stream_cut_over_two_chain_nodes() {
    const size_t n = 2 * 40960, half = n / 2;
    std::vector<C> x(n);
    for (size_t i = 0; i < n; ++i) {
        const double ph = -2.0 * M_PI * 0.05 * static_cast<double>(i) + 8.0 * std::cos(2.0 * M_PI * static_cast<double>(i) / 4096.0);
        x[i] = C(static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)));
    }
    std::vector<C> taps(127);
    for (int k = 0; k < 127; ++k) {
        const double t = k - 63.0, sc = t == 0.0 ? 1.0 : std::sin(M_PI * t / 8.0) / (M_PI * t / 8.0);
        taps[k] = C(static_cast<float>(0.125 * sc * (0.54 - 0.46 * std::cos(2.0 * M_PI * k / 126.0))), 0.f);
    }
    const double dphase = 2.0 * M_PI * 0.05, phase0 = 0.4;
    const size_t rate = 8;
    CHECK(shard_range(n, 2, 0) == std::make_pair(size_t(0), half) && shard_range(n, 2, 1) == std::make_pair(half, n));
    CHECK(shard_range(10, 4, 1) == std::make_pair(size_t(3), size_t(6)) && shard_range(10, 4, 3) == std::make_pair(size_t(8), size_t(10)));
    const size_t W = chain_prefix_len(taps.size(), rate, true);
    CHECK(W == 136 && chain_prefix_len(255, 8, false) == 256 && chain_prefix_len(1, 8, false) == 0);

    ChainNodeDev<float> whole(dphase, phase0, taps, rate);
    const std::vector<float> want = whole.run(DeviceBuf<C>::from_host(x)).value().to_host();

    ChainNodeDev<float> left(dphase, phase0, taps, rate);
    ChainNodeDev<float> right(dphase, shard_mixer_phase(phase0, dphase, static_cast<long long>(half - W)), taps, rate);
    std::vector<float> got = left.run(DeviceBuf<C>::from_host(std::vector<C>(x.begin(), x.begin() + half))).value().to_host();
    (void)right.run(DeviceBuf<C>::from_host(std::vector<C>(x.begin() + (half - W), x.begin() + half))).value();  // primed
    const std::vector<float> g1 = right.run(DeviceBuf<C>::from_host(std::vector<C>(x.begin() + half, x.end()))).value().to_host();
    got.insert(got.end(), g1.begin(), g1.end());
    CHECK(got.size() == want.size());
    double worst = 0.0;
    for (size_t i = 0; i < got.size() && i < want.size(); ++i) {
        double e = std::fabs(static_cast<double>(got[i]) - want[i]);
        if (e > M_PI) e = 2.0 * M_PI - e;
        worst = std::fmax(worst, e);
    }
    CHECK(worst < 1e-5);  // the oscillator restarts from a closed-form phase: not the same f32 everywhere, 1e-5 rad

    // node by node: FIR -> mixer -> decimate, the second shard's FIR state from the halo: as many samples as taps
    // (the reference's state holds taps.len() samples, fir_node.rs:97-105; fewer would truncate the filter, fir.rs:53)
    BatchFirNode f_all(taps);
    BatchMixerNode m_all(dphase, phase0);
    const std::vector<C> ref = DecimateNode<C>(rate).run(m_all.run(f_all.run(x).value()).value()).value();
    BatchFirNode f_r(taps, state_from_halo(std::vector<C>(x.begin() + (half - 127), x.begin() + half)));
    BatchMixerNode m_r(dphase, shard_mixer_phase(phase0, dphase, static_cast<long long>(half)));
    const std::vector<C> r2 =
        DecimateNode<C>(rate).run(m_r.run(f_r.run(std::vector<C>(x.begin() + half, x.end())).value()).value()).value();
    CHECK(r2.size() * 2 == ref.size());
    double w2 = 0.0;
    for (size_t i = 0; i < r2.size() && r2.size() * 2 == ref.size(); ++i) w2 = std::fmax(w2, std::abs(r2[i] - ref[r2.size() + i]));
    CHECK(w2 < 2e-6);
}

static void test_device_resident_chain_and_fft() {
    // config 3 as a graph of device-resident messages: source -> fused chain (mixer, 127-tap LPF, /8,
    // FM demod) -> sink, against the four host-vector nodes in series; then FFT -> IFFT round trip
    const size_t n = 8 * 8192;
    std::vector<C> x(n);
    for (size_t i = 0; i < n; ++i) {
        const double ph = -2.0 * M_PI * 0.05 * static_cast<double>(i) + 8.0 * std::cos(2.0 * M_PI * static_cast<double>(i) / 4096.0);
        x[i] = C(static_cast<float>(std::cos(ph)), static_cast<float>(std::sin(ph)));
    }
    std::vector<C> taps(127);
    for (int k = 0; k < 127; ++k) {
        const double t = k - 63.0, sc = t == 0.0 ? 1.0 : std::sin(M_PI * t / 8.0) / (M_PI * t / 8.0);
        taps[k] = C(static_cast<float>(0.125 * sc * (0.54 - 0.46 * std::cos(2.0 * M_PI * k / 126.0))), 0.f);
    }
    const double dphase = 2.0 * M_PI * 0.05;
    BatchMixerNode m(dphase);
    BatchFirNode f(taps);
    DecimateNode<C> d(8);
    FMDemodNode fm;
    auto want = fm.run(d.run(f.run(m.run(x).value()).value()).value()).value();

    Collect<DeviceBuf<float>> chk;
    Replay<DeviceBuf<C>> src({DeviceBuf<C>::from_host(std::vector<C>(x.begin(), x.begin() + n / 2)),
                              DeviceBuf<C>::from_host(std::vector<C>(x.begin() + n / 2, x.end()))});
    ChainNodeDev<float> chain(dphase, 0.0, taps, 8);
    CHECK(chain.fused_kind() == 2);
    connect_nodes(src.output, chain.input);
    connect_nodes(chain.output, chk.input);
    start_nodes(std::move(src), std::move(chain));
    while (chk.call().is_ok()) {
    }
    CHECK(chk.got.size() == 2);
    if (chk.got.size() == 2) {
        std::vector<float> got = chk.got[0].to_host();
        const std::vector<float> g1 = chk.got[1].to_host();
        got.insert(got.end(), g1.begin(), g1.end());
        CHECK(got.size() == want.size());
        double worst = 0.0;
        for (size_t i = 64; i < got.size() && i < want.size(); ++i) {  // past the filter's start-up transient
            double e = std::fabs(static_cast<double>(got[i]) - want[i]);
            if (e > M_PI) e = 2.0 * M_PI - e;
            worst = std::fmax(worst, e);
        }
        CHECK(worst < 1e-4);
    }

    FFTBatchNodeDev fwd(1024, false), inv(1024, true);
    DeviceBuf<C> X = fwd.run(DeviceBuf<C>::from_host(std::vector<C>(x.begin(), x.begin() + 16 * 1024))).value();
    std::vector<C> back = inv.run(X).value().to_host();
    double worst = 0.0;
    for (size_t i = 0; i < back.size(); ++i) worst = std::fmax(worst, std::abs(back[i] / 1024.0f - x[i]));
    CHECK(worst < 1e-5);

    // pulse shaping on a symbol block equals upsample + FIR (rect taps x4: every symbol held 4 samples)
    std::vector<C> sym(1000);
    for (size_t i = 0; i < sym.size(); ++i) sym[i] = C((i * 7) % 3 == 0 ? 1.f : -1.f, (i * 5) % 4 < 2 ? 1.f : -1.f);
    BatchPulseNodeDev shaper(rect_taps(4), 4);
    std::vector<C> shaped = shaper.run(DeviceBuf<C>::from_host(sym)).value().to_host();
    CHECK(shaped.size() == 4000);
    bool held = true;
    for (size_t i = 0; i < shaped.size(); ++i) held = held && shaped[i] == sym[i / 4];
    CHECK(held);
    // the transmit chain as one launch: the shaped block times the mixer's oscillator (phase 0.25 + 0.3 i)
    BatchPulseNodeDev tx(rect_taps(4), 4);
    tx.with_mixer(0.3, 0.25);
    std::vector<C> mixed = tx.run(DeviceBuf<C>::from_host(sym)).value().to_host();
    double mworst = 0.0;
    for (size_t i = 0; i < mixed.size(); ++i) {
        const double ph = 0.25 + 0.3 * static_cast<double>(i);
        const std::complex<double> want = std::complex<double>(sym[i / 4]) * std::complex<double>(std::cos(ph), std::sin(ph));
        mworst = std::fmax(mworst, std::abs(std::complex<double>(mixed[i]) - want));
    }
    CHECK(mworst < 1e-5);
    UpsampleNodeDev up(4);
    std::vector<C> stuffed = up.run(DeviceBuf<C>::from_host(sym)).value().to_host();
    bool zs = stuffed.size() == 4000;
    for (size_t i = 0; i < stuffed.size() && zs; ++i) zs = stuffed[i] == (i % 4 ? C(0, 0) : sym[i / 4]);
    CHECK(zs);
}

// Per-sample drop-in nodes must not regress an existing graph: FirNode / MixerNode receive single
// samples (src/filter/fir_node.rs:111-113, src/mixer.rs:145-147) and a launch per sample would cap the
// graph at ~40 ksamples/s.  DeriveNode::call drains what is queued and runs it as one launch: 1 M
// samples through source -> MixerNode -> FirNode -> sink, per-sample messages on every channel.
static void test_per_sample_nodes_keep_up() {
    const size_t n = 1u << 20;
    std::vector<C> x(n);
    comms_synth_iq_host(c32(x.data()), n, 0, 99);
    const auto taps = rrc_taps(63, 4.0, 0.25);
    Collect<C> chk;
    double best = 0.0;
    for (int attempt = 0; attempt < 1; ++attempt) {  // wall-clock rate of five host threads, printed
        chk = Collect<C>();
        chk.got.reserve(n);
        const auto t0 = std::chrono::steady_clock::now();
        {
            Replay<C> src(x);
            MixerNode mix(0.123, 0.1);
            FirNode fir(taps);
            connect_nodes(src.output, mix.input);
            connect_nodes(mix.output, fir.input);
            connect_nodes(fir.output, chk.input);
            start_nodes(std::move(src), std::move(mix), std::move(fir));
            while (chk.call().is_ok()) {
            }
        }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("per-sample graph MixerNode -> FirNode: %zu samples in %.3f s = %.2f Msamples/s\n", n, dt, n / dt / 1e6);
        best = std::max(best, n / dt);
    }
    CHECK(chk.got.size() == n);
    // (the rate is printed, not asserted: 12.9-20.6 Msamples/s over the boxes seen; a loaded host must not turn a
    // correctness test red -- ADVICE round 2)
    (void)best;
    // same samples through the batch forms in one call each: same stream, same state evolution
    BatchMixerNode bm(0.123, 0.1);
    BatchFirNode bf(taps);
    auto want = bf.run(bm.run(x).value()).value();
    double worst = 0.0;
    for (size_t i = 0; i < n && i < chk.got.size(); ++i) worst = std::max<double>(worst, std::abs(chk.got[i] - want[i]));
    double tsum = 0.0;
    for (auto& t : taps) tsum += std::abs(t);
    CHECK(worst <= 1e-5 * tsum * 1.5);   // |x| < 1.42; block cuts pick different FIR kernels (direct / overlap-save)
    // a trickle (one sample at a time, each awaited) still comes straight through
    {
        MixerNode mix(0.5);
        NodeSender<C> in;
        NodeReceiver<C> out;
        connect_nodes(in, mix.input);
        connect_nodes(mix.output, out);
        for (int i = 0; i < 3; ++i) {
            CHECK(in[0].first.send(C(1, 0)));
            CHECK(mix.call().is_ok());
            auto v = out->try_recv();
            CHECK(v.has_value() && close(*v, C(std::cos(0.5f * i), std::sin(0.5f * i)), 1e-6f));
        }
    }
}

int main() {
    int32_t ndev = 0;
    if (comms_device_count(&ndev) != COMMS_OK || ndev < 1) {
        std::fprintf(stderr, "no MI355X visible: %s\n", comms_last_error());
        return 2;
    }
    test_fir_nodes();
    test_integer_nodes();
    test_f64_nodes();
    test_fft_nodes();
    test_mixer_nodes();
    test_pulse_node();
    test_resample_nodes();
    test_fm_node();
    test_fm_radio_literal_graph();
    test_demod_nodes();
    test_device_resident_graph();
    test_device_resident_stream_many_messages();
    test_device_resident_fan_out();
    test_device_resident_chain_and_fft();
    test_stream_cut_over_two_chain_nodes();
    test_per_sample_nodes_keep_up();
    if (g_fail) {
        std::fprintf(stderr, "%d check(s) failed\n", g_fail);
        return 1;
    }
    std::puts("host GPU node tests: all passed");
    return 0;
}
