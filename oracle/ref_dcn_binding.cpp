// Binding shim for the COMPILED-REFERENCE DCN checker (oracle/_ref/_d2ref_C*.so).
// TEST INFRASTRUCTURE ONLY -- nothing under detectron2_amd/ may load it.  This file contains no reference
// code: it #includes the reference's own header where it lies and exposes the five header dispatchers
//   detectron2/layers/csrc/deformable/deform_conv.h:63-375
// under the names the reference's vision.cpp:86-102 gives them, so that the reference's own
// layers/deform_conv.py (_DeformConv / _ModulatedDeformConv, forward AND backward) runs on top of the
// reference's own kernels (deform_conv_cuda.cu, deform_conv_cuda_kernel.cu compiled as HIP for gfx950).
#include <torch/extension.h>
#include "deformable/deform_conv.h"

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  using namespace detectron2;
  m.def("deform_conv_forward", &deform_conv_forward);
  m.def("deform_conv_backward_input", &deform_conv_backward_input);
  m.def("deform_conv_backward_filter", &deform_conv_backward_filter);
  m.def("modulated_deform_conv_forward", &modulated_deform_conv_forward);
  m.def("modulated_deform_conv_backward", &modulated_deform_conv_backward);
}
