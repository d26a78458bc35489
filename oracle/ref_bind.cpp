// ref_bind.cpp — TEST INFRASTRUCTURE.  Python binding (ours) for the REFERENCE's own CPU eval
// helpers, which are compiled from where they lie:
//   /root/reference/mmdet3d_gaussian/ops/eval/affinity.cpp   (iou_3d :8-49, iou_bev :51-81, trans_bev :83-105)
//   /root/reference/mmdet3d_gaussian/ops/eval/rbox_utils.hpp (included by affinity.cpp)
//   /root/reference/mmdet3d_gaussian/ops/eval/matcher.cpp    (match_coco :8-74)
// The reference binds them in ops/eval/eval_utils.cpp:26-36 through torch/extension.h; this file
// declares the same symbols and binds them with plain pybind11 so that the build needs neither
// torch headers nor the reference's setup.py.  Output: oracle/_ref/ref_eval*.so (git-ignored).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

namespace py = pybind11;

namespace eval {
namespace matcher {
py::array_t<int32_t> match_coco(const py::array_t<float> &cost_mat_, const py::array_t<float> &cost_thrs_,
                                const py::array_t<bool> &is_ignore_, const py::array_t<bool> &is_crowd_);
}  // namespace matcher
namespace affinity {
py::array_t<float> iou_3d(const py::array_t<float> &det_, const py::array_t<float> &gt_, const float z_offset);
py::array_t<float> iou_bev(const py::array_t<float> &det_, const py::array_t<float> &gt_);
py::array_t<float> trans_bev(const py::array_t<float> &det_, const py::array_t<float> &gt_);
}  // namespace affinity
}  // namespace eval

PYBIND11_MODULE(ref_eval, m) {
  m.def("iou_3d", &eval::affinity::iou_3d, py::arg("det").noconvert(), py::arg("gt").noconvert(),
        py::arg("z_offset") = 0.5f);
  m.def("iou_bev", &eval::affinity::iou_bev, py::arg("det").noconvert(), py::arg("gt").noconvert());
  m.def("trans_bev", &eval::affinity::trans_bev, py::arg("det").noconvert(), py::arg("gt").noconvert());
  m.def("match_coco", &eval::matcher::match_coco, py::arg("cost_mat").noconvert(), py::arg("cost_thrs").noconvert(),
        py::arg("is_ignore").noconvert(), py::arg("is_crowd").noconvert());
}
