// torch_ext.cpp -- the PyTorch-ROCm extension host of the engine (BASELINE.json north_star: "server-python/FastAPI path
// calling the same engine via PyTorch-ROCm extension"; SURVEY.md 8(b) caller 2).
//
// A torch C++ extension (built in-tree by build.py with the toolchain torch.utils.cpp_extension describes) that takes and
// returns at::Tensor and calls the engine's C ABI (include/ire.h) with tensor.data_ptr() and the CURRENT torch HIP stream:
// torch owns device memory and streams, libire.so does the work.  Errors are raised as RuntimeError("[ire status N] message")
// -- the message keeps the engine's "invalid" / "timeout" / "service unavailable" wording (restorator.js:241-265).
//   classify(h, rgb[N,H,W,3] u8, is_jpeg[N] u8?)            -> (scores[N,7] f64, labels[N] i32)      ClassifierService.analyze
//   restore(h, rgb[N,H,W,3] u8, scores[N,7] f64?, is_jpeg?) -> restored[N,H,W,3] u8                  GeminiClient.restoreImage
//   fuse(h, views[k,H,W,3] u8, noise)                       -> (fused[H,W,3] u8, shifts[k,2] i32)    restoreImage, 2..3 images
//   restore_tiled(h, rgb[H,W,3] u8, nstrips, scores[7]?, is_jpeg[1]?) -> restored[H,W,3] u8           cfg 4 (row strips)
//   fuse_batch(h, views[S,k,H,W,3] u8, noise[S] f64 (cpu))  -> (fused[S,H,W,3] u8, shifts[S,k,2] i32) S <= 16 restoreImage calls, one kernel chain
//   preprocess(h, rgb[H,W,3] u8, orientation, max_dim)      -> upright, fitted [H',W',3] u8          imagePreprocess.js:24-91 (pixel part)
//   encode_png_base64(h, rgb[N,H,W,3] u8)                   -> chars[N, ire_png_base64_bytes] u8     restorator.js:108 (the result text, on the device)
#include <c10/hip/HIPStream.h>
#include <dlfcn.h>
#include <torch/extension.h>

#include <stdexcept>
#include <string>

#include "../../include/ire.h"

namespace {

struct Api {
    void* so = nullptr;
    decltype(&ire_init) init = nullptr;
    decltype(&ire_shutdown) shutdown = nullptr;
    decltype(&ire_last_error) last_error = nullptr;
    decltype(&ire_abi_version) abi_version = nullptr;
    decltype(&ire_classify_device) classify_device = nullptr;
    decltype(&ire_restore_device) restore_device = nullptr;
    decltype(&ire_fuse_device) fuse_device = nullptr;
    decltype(&ire_restore_tiled_device) restore_tiled_device = nullptr;
    decltype(&ire_fuse_batch_device) fuse_batch_device = nullptr;
    decltype(&ire_preprocess_plan) preprocess_plan = nullptr;
    decltype(&ire_preprocess_device) preprocess_device = nullptr;
    decltype(&ire_png_base64_bytes) png_base64_bytes = nullptr;
    decltype(&ire_encode_png_base64_device) encode_png_base64_device = nullptr;
} g;

void load(const std::string& path) {
    if (g.so) return;
    void* so = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
    if (!so) throw std::runtime_error(std::string("[ire status 3] service unavailable: cannot load ") + path + ": " + dlerror());
#define SYM(f, n) g.f = (decltype(g.f))dlsym(so, n); if (!g.f) throw std::runtime_error("[ire status 3] service unavailable: missing symbol " n);
    SYM(init, "ire_init") SYM(shutdown, "ire_shutdown") SYM(last_error, "ire_last_error") SYM(abi_version, "ire_abi_version")
    SYM(classify_device, "ire_classify_device") SYM(restore_device, "ire_restore_device") SYM(fuse_device, "ire_fuse_device")
    SYM(restore_tiled_device, "ire_restore_tiled_device") SYM(fuse_batch_device, "ire_fuse_batch_device")
    SYM(preprocess_plan, "ire_preprocess_plan") SYM(preprocess_device, "ire_preprocess_device")
    SYM(png_base64_bytes, "ire_png_base64_bytes") SYM(encode_png_base64_device, "ire_encode_png_base64_device")
#undef SYM
    if (g.abi_version() != IRE_ABI_VERSION) throw std::runtime_error("[ire status 3] service unavailable: libire.so ABI version mismatch");
    g.so = so;
}

void check(int rc) {
    if (rc != 0) throw std::runtime_error("[ire status " + std::to_string(rc) + "] " + g.last_error());
}

ire_engine* eng(int64_t h) {
    if (!h) throw std::runtime_error("[ire status 1] invalid engine handle");
    return reinterpret_cast<ire_engine*>(h);
}

void want(const at::Tensor& t, at::ScalarType ty, int64_t dim, const char* what) {
    TORCH_CHECK(t.is_cuda() && t.scalar_type() == ty && t.is_contiguous() && (dim < 0 || t.dim() == dim), "[ire status 1] invalid input: ", what);
}
void* stream_of(const at::Tensor& t) { return (void*)c10::hip::getCurrentHIPStream(t.get_device()).stream(); }

int64_t init(const std::string& lib_path, int64_t device_index, int64_t max_batch, int64_t num_streams, const std::string& weights_path,
             const std::string& precision) {
    load(lib_path);
    ire_config cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.device_index = (int32_t)device_index;
    cfg.precision = precision == "fp8" ? IRE_PRECISION_FP8 : IRE_PRECISION_BF16;
    cfg.max_batch = (int32_t)max_batch;
    cfg.num_streams = (int32_t)num_streams;
    cfg.weights_path = weights_path.empty() ? nullptr : weights_path.c_str();
    ire_engine* e = nullptr;
    check(g.init(&cfg, &e));
    return reinterpret_cast<int64_t>(e);
}

void shutdown(int64_t h) { if (h && g.shutdown) g.shutdown(reinterpret_cast<ire_engine*>(h)); }

std::tuple<at::Tensor, at::Tensor> classify(int64_t h, const at::Tensor& rgb, const c10::optional<at::Tensor>& is_jpeg) {
    want(rgb, at::kByte, 4, "rgb must be a contiguous cuda uint8 [N,H,W,3]");
    if (is_jpeg) want(*is_jpeg, at::kByte, 1, "is_jpeg must be a cuda uint8 [N]");
    const int n = (int)rgb.size(0), hh = (int)rgb.size(1), ww = (int)rgb.size(2);
    at::Tensor scores = at::empty({n, 7}, rgb.options().dtype(at::kDouble));
    at::Tensor labels = at::empty({n}, rgb.options().dtype(at::kInt));
    check(g.classify_device(eng(h), rgb.data_ptr<uint8_t>(), n, hh, ww, is_jpeg ? is_jpeg->data_ptr<uint8_t>() : nullptr, scores.data_ptr<double>(),
                            labels.data_ptr<int32_t>(), stream_of(rgb)));
    return {scores, labels};
}

at::Tensor restore(int64_t h, const at::Tensor& rgb, const c10::optional<at::Tensor>& scores, const c10::optional<at::Tensor>& is_jpeg) {
    want(rgb, at::kByte, 4, "rgb must be a contiguous cuda uint8 [N,H,W,3]");
    if (scores) want(*scores, at::kDouble, 2, "scores must be a cuda float64 [N,7]");
    if (is_jpeg) want(*is_jpeg, at::kByte, 1, "is_jpeg must be a cuda uint8 [N]");
    at::Tensor out = at::empty_like(rgb);
    check(g.restore_device(eng(h), rgb.data_ptr<uint8_t>(), (int)rgb.size(0), (int)rgb.size(1), (int)rgb.size(2),
                           scores ? scores->data_ptr<double>() : nullptr, is_jpeg ? is_jpeg->data_ptr<uint8_t>() : nullptr, out.data_ptr<uint8_t>(),
                           stream_of(rgb)));
    return out;
}

std::tuple<at::Tensor, at::Tensor> fuse(int64_t h, const at::Tensor& views, double noise) {
    want(views, at::kByte, 4, "views must be a contiguous cuda uint8 [k,H,W,3]");
    const int k = (int)views.size(0), hh = (int)views.size(1), ww = (int)views.size(2);
    at::Tensor out = at::empty({hh, ww, 3}, views.options());
    at::Tensor shifts = at::zeros({k, 2}, views.options().dtype(at::kInt));
    check(g.fuse_device(eng(h), views.data_ptr<uint8_t>(), k, hh, ww, noise, out.data_ptr<uint8_t>(), shifts.data_ptr<int32_t>(), stream_of(views)));
    return {out, shifts};
}

at::Tensor restore_tiled(int64_t h, const at::Tensor& rgb, int64_t nstrips, const c10::optional<at::Tensor>& scores,
                         const c10::optional<at::Tensor>& is_jpeg) {
    want(rgb, at::kByte, 3, "rgb must be a contiguous cuda uint8 [H,W,3]");
    if (scores) want(*scores, at::kDouble, -1, "scores must be a cuda float64 [7]");
    at::Tensor out = at::empty_like(rgb);
    check(g.restore_tiled_device(eng(h), rgb.data_ptr<uint8_t>(), (int)rgb.size(0), (int)rgb.size(1), (int)nstrips,
                                 scores ? scores->data_ptr<double>() : nullptr, is_jpeg ? is_jpeg->data_ptr<uint8_t>() : nullptr, out.data_ptr<uint8_t>(),
                                 stream_of(rgb)));
    return out;
}

std::tuple<at::Tensor, at::Tensor> fuse_batch(int64_t h, const at::Tensor& views, const at::Tensor& noise) {
    want(views, at::kByte, 5, "views must be a contiguous cuda uint8 [S,k,H,W,3]");
    TORCH_CHECK(!noise.is_cuda() && noise.scalar_type() == at::kDouble && noise.is_contiguous() && noise.dim() == 1 && noise.size(0) == views.size(0),
                "[ire status 1] invalid input: noise must be a cpu float64 [S] (< 0: classify view 0 of that set inside)");
    const int s = (int)views.size(0), k = (int)views.size(1), hh = (int)views.size(2), ww = (int)views.size(3);
    at::Tensor out = at::empty({s, hh, ww, 3}, views.options());
    at::Tensor shifts = at::zeros({s, k, 2}, views.options().dtype(at::kInt));
    check(g.fuse_batch_device(eng(h), views.data_ptr<uint8_t>(), s, k, hh, ww, noise.data_ptr<double>(), out.data_ptr<uint8_t>(),
                              shifts.data_ptr<int32_t>(), stream_of(views)));
    return {out, shifts};
}

at::Tensor preprocess(int64_t h, const at::Tensor& rgb, int64_t orientation, int64_t max_dim) {
    want(rgb, at::kByte, 3, "rgb must be a contiguous cuda uint8 [H,W,3]");
    const int hh = (int)rgb.size(0), ww = (int)rgb.size(1);
    int ow = 0, oh = 0, resized = 0;
    check(g.preprocess_plan(ww, hh, (int)orientation, (int)max_dim, &ow, &oh, &resized));
    at::Tensor out = at::empty({oh, ow, 3}, rgb.options());
    check(g.preprocess_device(eng(h), rgb.data_ptr<uint8_t>(), hh, ww, (int)orientation, (int)max_dim, out.data_ptr<uint8_t>(), oh, ow, stream_of(rgb)));
    return out;
}

at::Tensor encode_png_base64(int64_t h, const at::Tensor& rgb) {
    want(rgb, at::kByte, 4, "rgb must be a contiguous cuda uint8 [N,H,W,3]");
    const int n = (int)rgb.size(0), hh = (int)rgb.size(1), ww = (int)rgb.size(2);
    const size_t cb = g.png_base64_bytes(hh, ww);
    TORCH_CHECK(cb != 0, "[ire status 1] invalid image size for the PNG encoder: width must be a multiple of 8");
    at::Tensor out = at::empty({n, (int64_t)cb}, rgb.options());
    check(g.encode_png_base64_device(eng(h), rgb.data_ptr<uint8_t>(), n, hh, ww, out.data_ptr<uint8_t>(), cb, stream_of(rgb)));
    return out;
}

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    m.doc() = "PyTorch-ROCm extension host of the MI355X image-restoration engine (libire.so, include/ire.h)";
    m.def("init", &init, "create an engine; returns its handle");
    m.def("shutdown", &shutdown);
    m.def("classify", &classify, py::arg("handle"), py::arg("rgb"), py::arg("is_jpeg") = py::none());
    m.def("restore", &restore, py::arg("handle"), py::arg("rgb"), py::arg("scores") = py::none(), py::arg("is_jpeg") = py::none());
    m.def("fuse", &fuse, py::arg("handle"), py::arg("views"), py::arg("noise") = -1.0);
    m.def("fuse_batch", &fuse_batch, py::arg("handle"), py::arg("views"), py::arg("noise"));
    m.def("preprocess", &preprocess, py::arg("handle"), py::arg("rgb"), py::arg("orientation") = 1, py::arg("max_dim") = 2048);
    m.def("encode_png_base64", &encode_png_base64, py::arg("handle"), py::arg("rgb"));
    m.def("restore_tiled", &restore_tiled, py::arg("handle"), py::arg("rgb"), py::arg("nstrips"), py::arg("scores") = py::none(),
          py::arg("is_jpeg") = py::none());
}
