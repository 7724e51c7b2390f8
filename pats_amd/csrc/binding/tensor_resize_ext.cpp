// The reference's one native boundary, compiled: Python extension module `tensor_resize` with the single function
// `tensor_resize(input, bound) -> Tensor` (reference: setup/library.cpp:47-66 `torch::Tensor resize(...)`, module
// definition :92-93, built by setup/setup.py:107-118, imported at utils/utils.py:17, called at utils/utils.py:1385).
//
// This file is the libtorch / pybind11 side only: it checks the tensors, allocates the result, and hands raw device
// pointers + the caller's current HIP stream to the C-ABI `pats_tensor_resize_f32` of libpats_amd.so
// (include/pats_amd.h), where the gather kernel lives (csrc/resize.hip).  There is no host path: a CPU tensor raises.
#include <torch/extension.h>
// PyTorch-ROCm keeps the device type named `cuda` (torch.device("cuda") IS the MI355X), so the guard and the stream
// come from the HIP implementations registered under that name; the plain c10::hip::HIPGuard refuses such a device.
#include <ATen/hip/impl/HIPGuardImplMasqueradingAsCUDA.h>
#include <ATen/hip/impl/HIPStreamMasqueradingAsCUDA.h>

#include "pats_amd.h"

namespace {

constexpr int64_t kOut = 96;      // library.cpp:50-51: patch_shape * 3

torch::Tensor resize(const torch::Tensor& input_tensor, const torch::Tensor& bound) {
  TORCH_CHECK(input_tensor.is_cuda(),
              "tensor_resize: input must live on the GPU (pats_amd has no CPU fallback), got ", input_tensor.device());
  TORCH_CHECK(bound.device() == input_tensor.device(), "tensor_resize: bound is on ", bound.device(),
              ", input on ", input_tensor.device());
  TORCH_CHECK(input_tensor.scalar_type() == torch::kFloat, "tensor_resize: input must be float32, got ",
              input_tensor.scalar_type());
  TORCH_CHECK(bound.scalar_type() == torch::kLong, "tensor_resize: bound must be int64, got ", bound.scalar_type());
  TORCH_CHECK(input_tensor.dim() == 4, "tensor_resize: input must be [n,C,H,W], got ", input_tensor.dim(), " dims");
  TORCH_CHECK(bound.dim() == 2 && bound.size(1) == 5, "tensor_resize: bound must be [K,5] (y0,y1,x0,x1,seq)");
  TORCH_CHECK(input_tensor.size(0) < (1 << 30) && input_tensor.size(1) < (1 << 30) &&
              input_tensor.size(2) < (1 << 30) && input_tensor.size(3) < (1 << 30), "tensor_resize: input too large");

  const c10::hip::HIPGuardMasqueradingAsCUDA guard(input_tensor.device());
  const auto in = input_tensor.contiguous();      // borrowed; a strided view is packed, never written
  const auto bnd = bound.contiguous();
  const int64_t K = bnd.size(0);
  auto out = torch::empty({K, in.size(1), kOut, kOut}, in.options());
  if (K == 0) return out;                          // nothing matched: the reference's loop body never runs

  // library.cpp:56-60: narrow() beyond the tensor and upsample of an empty crop are c10::Error there.  The kernel is
  // memory-safe for any bound (it clamps) and raises this flag instead; reading it back is the call's one host sync
  // (the reference makes five per crop).
  auto status = torch::zeros({1}, bnd.options().dtype(torch::kInt));
  const auto stream = c10::hip::getCurrentHIPStreamMasqueradingAsCUDA(input_tensor.device().index());
  const int rc = pats_tensor_resize_f32(in.data_ptr<float>(), static_cast<int>(in.size(0)), static_cast<int>(in.size(1)),
                                        static_cast<int>(in.size(2)), static_cast<int>(in.size(3)),
                                        bnd.data_ptr<int64_t>(), K, out.data_ptr<float>(), status.data_ptr<int32_t>(),
                                        static_cast<pats_stream_t>(stream.stream()));
  TORCH_CHECK(rc == 0, "tensor_resize: ", pats_last_error());
  TORCH_CHECK(status.item<int32_t>() == 0, "tensor_resize: a crop is empty or outside the ", in.size(2), "x", in.size(3),
              " input (start/length out of range, or image index >= ", in.size(0), ")");
  return out;
}

}  // namespace

PYBIND11_MODULE(tensor_resize, m) {
  m.doc() = "pats_amd: MI355X drop-in for the reference's tensor_resize extension (setup/library.cpp)";
  m.def("tensor_resize", &resize, "feature resize", pybind11::arg("input_tensor"), pybind11::arg("bound"));
}
