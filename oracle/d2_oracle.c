/*
 * d2_oracle.c -- CPU ORACLE for the detectron2 per-image detection hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the smoke check in
 * __graft_entry__.py and bench.py's `cpu_baseline` leg may load this library, and only
 * as the checker.  The product path (detectron2_amd/) never imports, links or calls it.
 *
 * It is a plain-C restatement (fp32, single-threaded, no FMA contraction -- build with
 * -ffp-contract=off) of the reference algorithms.  All citations are relative to
 * /root/reference/.  Where the arithmetic lives in torchvision (not vendored by the
 * reference, no version pin: INSTALL.md:5 "matches the PyTorch installation"; CI uses
 * 0.14.1/0.17.2/0.19.1, .github/workflows/workflow.yml:43-48), the function restates
 * torchvision's published CPU algorithm and is anchored on the reference's own call
 * sites, tests and golden vectors (tests/golden/README.md lists them).
 *
 * Parity pinning status (see tests/test_oracle_golden.py):
 *   roi_align fwd/bwd          pinned: known answers tests/layers/test_roi_align.py:14-47,
 *                              == ROIAlignRotated(0 deg) (tests/modeling/test_roi_pooler.py:14-59)
 *   roi_align_rotated fwd/bwd  pinned: compiled reference (oracle/_ref) + goldens
 *   box_iou_rotated, nms_rotated  pinned: compiled reference + known answers
 *   pairwise_iou/ioa/intersection pinned: reference python (structures/boxes.py) + goldens
 *   paste_masks                pinned: reference python (layers/mask_ops.py) fixtures
 *   nms / batched_nms          pinned: reference's greedy oracle
 *                              tests/layers/test_nms_rotated.py:44-66; threshold tie / sort tie
 *                              behaviour is parity-unpinned by the reference's tests.
 *   deform_conv v1/v2 fwd      pinned: golden 5x5 tests/layers/test_deformable.py:38-58
 *   deform_conv v1/v2 bwd      PARITY UNPINNED (no CPU implementation or gradient test in the
 *                              reference); checked by fp64 finite differences instead.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------ */
/* ROIAlign (axis-aligned).  Call site: detectron2/layers/roi_align.py:58-65 ->
 * torchvision.ops.roi_align.  Restates torchvision's CPU kernel
 * (torchvision/csrc/ops/cpu/roi_align_kernel.cpp + roi_align_common.h), whose in-tree twin is
 * detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp:27-129 (pre_calc) and
 * :201-310 (forward) at theta = 0.  Pixel model: roi_align.py:15-35. */

typedef struct {
  int pos1, pos2, pos3, pos4;
  float w1, w2, w3, w4;
} orc_precalc;

/* ROIAlignRotated_cpu.cpp:64-125 (boundary handling shared by both variants) */
static void orc_bilinear_precalc(int height, int width, float y, float x, orc_precalc* pc) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) {
    memset(pc, 0, sizeof(*pc));
    return;
  }
  if (y < 0) y = 0;
  if (x < 0) x = 0;
  int y_low = (int)y;
  int x_low = (int)x;
  int y_high, x_high;
  if (y_low >= height - 1) {
    y_high = y_low = height - 1;
    y = (float)y_low;
  } else {
    y_high = y_low + 1;
  }
  if (x_low >= width - 1) {
    x_high = x_low = width - 1;
    x = (float)x_low;
  } else {
    x_high = x_low + 1;
  }
  float ly = y - y_low;
  float lx = x - x_low;
  float hy = (float)(1. - ly), hx = (float)(1. - lx);
  pc->w1 = hy * hx; pc->w2 = hy * lx; pc->w3 = ly * hx; pc->w4 = ly * lx;
  pc->pos1 = y_low * width + x_low;
  pc->pos2 = y_low * width + x_high;
  pc->pos3 = y_high * width + x_low;
  pc->pos4 = y_high * width + x_high;
}

/* ROIAlignRotated_cpu.cpp:131-194 (bilinear_interpolate_gradient) */
static void orc_bilinear_grad(int height, int width, float y, float x, float* w1, float* w2,
                              float* w3, float* w4, int* x_low, int* x_high, int* y_low,
                              int* y_high) {
  if (y < -1.0 || y > height || x < -1.0 || x > width) {
    *w1 = *w2 = *w3 = *w4 = 0.f;
    *x_low = *x_high = *y_low = *y_high = -1;
    return;
  }
  if (y < 0) y = 0;
  if (x < 0) x = 0;
  *y_low = (int)y;
  *x_low = (int)x;
  if (*y_low >= height - 1) {
    *y_high = *y_low = height - 1;
    y = (float)*y_low;
  } else {
    *y_high = *y_low + 1;
  }
  if (*x_low >= width - 1) {
    *x_high = *x_low = width - 1;
    x = (float)*x_low;
  } else {
    *x_high = *x_low + 1;
  }
  float ly = y - *y_low;
  float lx = x - *x_low;
  float hy = (float)(1. - ly), hx = (float)(1. - lx);
  *w1 = hy * hx; *w2 = hy * lx; *w3 = ly * hx; *w4 = ly * lx;
}

static int orc_imax(int a, int b) { return a > b ? a : b; }

int orc_roi_align_forward(const float* input, const float* rois, int num_rois, int channels,
                          int height, int width, int pooled_h, int pooled_w,
                          float spatial_scale, int sampling_ratio, int aligned, float* output) {
  for (int n = 0; n < num_rois; n++) {
    const float* r = rois + n * 5;
    int batch_ind = (int)r[0];
    float offset = aligned ? 0.5f : 0.0f;
    float roi_start_w = r[1] * spatial_scale - offset;
    float roi_start_h = r[2] * spatial_scale - offset;
    float roi_end_w = r[3] * spatial_scale - offset;
    float roi_end_h = r[4] * spatial_scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (!aligned) { /* legacy: force malformed ROIs to be 1x1 */
      roi_width = roi_width > 1.f ? roi_width : 1.f;
      roi_height = roi_height > 1.f ? roi_height : 1.f;
    }
    float bin_size_h = roi_height / (float)pooled_h;
    float bin_size_w = roi_width / (float)pooled_w;
    int grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / pooled_h);
    int grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_w);
    const float count = (float)orc_imax(grid_h * grid_w, 1);
    int ntab = orc_imax(grid_h, 0) * orc_imax(grid_w, 0) * pooled_h * pooled_w;
    orc_precalc* tab = (orc_precalc*)malloc(sizeof(orc_precalc) * (size_t)orc_imax(ntab, 1));
    int t = 0;
    for (int ph = 0; ph < pooled_h; ph++)
      for (int pw = 0; pw < pooled_w; pw++)
        for (int iy = 0; iy < grid_h; iy++) {
          const float yy = roi_start_h + ph * bin_size_h +
              (float)(iy + .5f) * bin_size_h / (float)grid_h;
          for (int ix = 0; ix < grid_w; ix++) {
            const float xx = roi_start_w + pw * bin_size_w +
                (float)(ix + .5f) * bin_size_w / (float)grid_w;
            orc_bilinear_precalc(height, width, yy, xx, &tab[t++]);
          }
        }
    for (int c = 0; c < channels; c++) {
      const float* in = input + ((size_t)batch_ind * channels + c) * height * width;
      float* out = output + ((size_t)n * channels + c) * pooled_h * pooled_w;
      t = 0;
      for (int ph = 0; ph < pooled_h; ph++)
        for (int pw = 0; pw < pooled_w; pw++) {
          float v = 0.f;
          for (int iy = 0; iy < grid_h; iy++)
            for (int ix = 0; ix < grid_w; ix++) {
              orc_precalc pc = tab[t++];
              v += pc.w1 * in[pc.pos1] + pc.w2 * in[pc.pos2] + pc.w3 * in[pc.pos3] +
                  pc.w4 * in[pc.pos4];
            }
          v /= count;
          out[ph * pooled_w + pw] = v;
        }
    }
    free(tab);
  }
  return 0;
}

/* torchvision roi_align backward (CPU); in-tree twin ROIAlignRotated_cpu.cpp:312-416.
 * grad_input must be zero-filled by the caller (at::zeros in the reference). */
int orc_roi_align_backward(const float* grad_output, const float* rois, int num_rois,
                           int channels, int height, int width, int pooled_h, int pooled_w,
                           float spatial_scale, int sampling_ratio, int aligned,
                           float* grad_input) {
  for (int n = 0; n < num_rois; n++) {
    const float* r = rois + n * 5;
    int batch_ind = (int)r[0];
    float offset = aligned ? 0.5f : 0.0f;
    float roi_start_w = r[1] * spatial_scale - offset;
    float roi_start_h = r[2] * spatial_scale - offset;
    float roi_end_w = r[3] * spatial_scale - offset;
    float roi_end_h = r[4] * spatial_scale - offset;
    float roi_width = roi_end_w - roi_start_w;
    float roi_height = roi_end_h - roi_start_h;
    if (!aligned) {
      roi_width = roi_width > 1.f ? roi_width : 1.f;
      roi_height = roi_height > 1.f ? roi_height : 1.f;
    }
    float bin_size_h = roi_height / (float)pooled_h;
    float bin_size_w = roi_width / (float)pooled_w;
    int grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / pooled_h);
    int grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_w);
    const float count = (float)(grid_h * grid_w);
    for (int c = 0; c < channels; c++) {
      float* gin = grad_input + ((size_t)batch_ind * channels + c) * height * width;
      const float* gout = grad_output + ((size_t)n * channels + c) * pooled_h * pooled_w;
      for (int ph = 0; ph < pooled_h; ph++)
        for (int pw = 0; pw < pooled_w; pw++) {
          const float g = gout[ph * pooled_w + pw];
          for (int iy = 0; iy < grid_h; iy++) {
            const float y = roi_start_h + ph * bin_size_h +
                (float)(iy + .5f) * bin_size_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ix++) {
              const float x = roi_start_w + pw * bin_size_w +
                  (float)(ix + .5f) * bin_size_w / (float)grid_w;
              float w1, w2, w3, w4;
              int xl, xh, yl, yh;
              orc_bilinear_grad(height, width, y, x, &w1, &w2, &w3, &w4, &xl, &xh, &yl, &yh);
              float g1 = g * w1 / count, g2 = g * w2 / count;
              float g3 = g * w3 / count, g4 = g * w4 / count;
              if (xl >= 0 && xh >= 0 && yl >= 0 && yh >= 0) {
                gin[yl * width + xl] += g1;
                gin[yl * width + xh] += g2;
                gin[yh * width + xl] += g3;
                gin[yh * width + xh] += g4;
              }
            }
          }
        }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* ROIAlignRotated.  detectron2/layers/csrc/ROIAlignRotated/ROIAlignRotated_cpu.cpp:201-310
 * (forward) and :312-416 (backward), T = float.  rois are (K,6) [b, cx, cy, w, h, deg].
 * trig_mode selects how `cos(theta)` with a float argument is evaluated (:233-234): 0 = cosf/
 * sinf (the std::cos(float) overload), 1 = (float)cos((double)theta).  The compiled reference
 * decides which one is right; tests/test_oracle_golden.py pins it.  Returns -1 if a ROI has
 * negative size (AT_ASSERTM at :236-238). */
static void orc_rot_params(const float* r, float spatial_scale, int trig_mode, float* cw,
                           float* ch, float* rw, float* rh, float* cs, float* sn) {
  float offset = 0.5f;
  *cw = r[1] * spatial_scale - offset;
  *ch = r[2] * spatial_scale - offset;
  *rw = r[3] * spatial_scale;
  *rh = r[4] * spatial_scale;
  float theta = (float)(r[5] * M_PI / 180.0);
  if (trig_mode == 0) {
    *cs = cosf(theta);
    *sn = sinf(theta);
  } else {
    *cs = (float)cos((double)theta);
    *sn = (float)sin((double)theta);
  }
}

int orc_roi_align_rotated_forward(const float* input, const float* rois, int num_rois,
                                  int channels, int height, int width, int pooled_h,
                                  int pooled_w, float spatial_scale, int sampling_ratio,
                                  int trig_mode, float* output) {
  for (int n = 0; n < num_rois; n++) {
    const float* r = rois + n * 6;
    int batch_ind = (int)r[0];
    float cw, chh, roi_width, roi_height, cs, sn;
    orc_rot_params(r, spatial_scale, trig_mode, &cw, &chh, &roi_width, &roi_height, &cs, &sn);
    if (!(roi_width >= 0 && roi_height >= 0)) return -1;
    float bin_size_h = roi_height / (float)pooled_h;
    float bin_size_w = roi_width / (float)pooled_w;
    int grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / pooled_h);
    int grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_w);
    const float count = (float)orc_imax(grid_h * grid_w, 1);
    float roi_start_h = (float)(-roi_height / 2.0);
    float roi_start_w = (float)(-roi_width / 2.0);
    int ntab = orc_imax(grid_h, 0) * orc_imax(grid_w, 0) * pooled_h * pooled_w;
    orc_precalc* tab = (orc_precalc*)malloc(sizeof(orc_precalc) * (size_t)orc_imax(ntab, 1));
    int t = 0;
    for (int ph = 0; ph < pooled_h; ph++)
      for (int pw = 0; pw < pooled_w; pw++)
        for (int iy = 0; iy < grid_h; iy++) {
          const float yy = roi_start_h + ph * bin_size_h +
              (float)(iy + .5f) * bin_size_h / (float)grid_h;
          for (int ix = 0; ix < grid_w; ix++) {
            const float xx = roi_start_w + pw * bin_size_w +
                (float)(ix + .5f) * bin_size_w / (float)grid_w;
            float y = yy * cs - xx * sn + chh;
            float x = yy * sn + xx * cs + cw;
            orc_bilinear_precalc(height, width, y, x, &tab[t++]);
          }
        }
    for (int c = 0; c < channels; c++) {
      const float* in = input + ((size_t)batch_ind * channels + c) * height * width;
      float* out = output + ((size_t)n * channels + c) * pooled_h * pooled_w;
      t = 0;
      for (int ph = 0; ph < pooled_h; ph++)
        for (int pw = 0; pw < pooled_w; pw++) {
          float v = 0.f;
          for (int iy = 0; iy < grid_h; iy++)
            for (int ix = 0; ix < grid_w; ix++) {
              orc_precalc pc = tab[t++];
              v += pc.w1 * in[pc.pos1] + pc.w2 * in[pc.pos2] + pc.w3 * in[pc.pos3] +
                  pc.w4 * in[pc.pos4];
            }
          v /= count;
          out[ph * pooled_w + pw] = v;
        }
    }
    free(tab);
  }
  return 0;
}

int orc_roi_align_rotated_backward(const float* grad_output, const float* rois, int num_rois,
                                   int channels, int height, int width, int pooled_h,
                                   int pooled_w, float spatial_scale, int sampling_ratio,
                                   int trig_mode, float* grad_input) {
  for (int n = 0; n < num_rois; n++) {
    const float* r = rois + n * 6;
    int batch_ind = (int)r[0];
    float cw, chh, roi_width, roi_height, cs, sn;
    orc_rot_params(r, spatial_scale, trig_mode, &cw, &chh, &roi_width, &roi_height, &cs, &sn);
    if (!(roi_width >= 0 && roi_height >= 0)) return -1;
    float bin_size_h = roi_height / (float)pooled_h;
    float bin_size_w = roi_width / (float)pooled_w;
    int grid_h = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_height / pooled_h);
    int grid_w = (sampling_ratio > 0) ? sampling_ratio : (int)ceilf(roi_width / pooled_w);
    float roi_start_h = (float)(-roi_height / 2.0);
    float roi_start_w = (float)(-roi_width / 2.0);
    const float count = (float)(grid_h * grid_w);
    for (int c = 0; c < channels; c++) {
      float* gin = grad_input + ((size_t)batch_ind * channels + c) * height * width;
      const float* gout = grad_output + ((size_t)n * channels + c) * pooled_h * pooled_w;
      for (int ph = 0; ph < pooled_h; ph++)
        for (int pw = 0; pw < pooled_w; pw++) {
          const float g = gout[ph * pooled_w + pw];
          for (int iy = 0; iy < grid_h; iy++) {
            const float yy = roi_start_h + ph * bin_size_h +
                (float)(iy + .5f) * bin_size_h / (float)grid_h;
            for (int ix = 0; ix < grid_w; ix++) {
              const float xx = roi_start_w + pw * bin_size_w +
                  (float)(ix + .5f) * bin_size_w / (float)grid_w;
              float y = yy * cs - xx * sn + chh;
              float x = yy * sn + xx * cs + cw;
              float w1, w2, w3, w4;
              int xl, xh, yl, yh;
              orc_bilinear_grad(height, width, y, x, &w1, &w2, &w3, &w4, &xl, &xh, &yl, &yh);
              float g1 = g * w1 / count, g2 = g * w2 / count;
              float g3 = g * w3 / count, g4 = g * w4 / count;
              if (xl >= 0 && xh >= 0 && yl >= 0 && yh >= 0) {
                gin[yl * width + xl] += g1;
                gin[yl * width + xh] += g2;
                gin[yh * width + xl] += g3;
                gin[yh * width + xh] += g4;
              }
            }
          }
        }
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Axis-aligned pairwise IoU / IoA / intersection.  detectron2/structures/boxes.py:312-377.
 * torch.min/max propagate NaN; clamp keeps NaN; `inter > 0` is false for NaN -> 0. */
static float orc_tmin(float a, float b) { return (isnan(a) || isnan(b)) ? NAN : (a < b ? a : b); }
static float orc_tmax(float a, float b) { return (isnan(a) || isnan(b)) ? NAN : (a > b ? a : b); }

/* mode: 0 = iou (boxes.py:336-358), 1 = ioa (:361-377), 2 = intersection (:312-331) */
int orc_pairwise_iou(const float* b1, int n, const float* b2, int m, int mode, float* out) {
  for (int i = 0; i < n; i++) {
    const float* a = b1 + 4 * i;
    float area1 = (a[2] - a[0]) * (a[3] - a[1]);
    for (int j = 0; j < m; j++) {
      const float* b = b2 + 4 * j;
      float w = orc_tmin(a[2], b[2]) - orc_tmax(a[0], b[0]);
      float h = orc_tmin(a[3], b[3]) - orc_tmax(a[1], b[1]);
      if (w < 0) w = 0; /* clamp_(min=0): NaN stays NaN */
      if (h < 0) h = 0;
      float inter = w * h;
      float r;
      if (mode == 2) {
        r = inter;
      } else {
        float area2 = (b[2] - b[0]) * (b[3] - b[1]);
        if (inter > 0)
          r = (mode == 0) ? inter / (area1 + area2 - inter) : inter / area2;
        else
          r = 0.f;
      }
      out[(size_t)i * m + j] = r;
    }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Rotated IoU.  detectron2/layers/csrc/box_iou_rotated/box_iou_rotated_utils.h:59-390 with
 * T = float.  Mixed float/double promotions of the C++ are kept expression by expression. */
typedef struct { float x, y; } orc_pt;

static float orc_cross(orc_pt a, orc_pt b) { return a.x * b.y - b.x * a.y; }  /* :54-57 */
static float orc_dot(orc_pt a, orc_pt b) { return a.x * b.x + a.y * b.y; }    /* :48-50 */
static orc_pt orc_sub(orc_pt a, orc_pt b) { orc_pt r = {a.x - b.x, a.y - b.y}; return r; }

static void orc_rot_vertices(float xc, float yc, float w, float h, float a, orc_pt* pts) {
  /* :59-76 */
  double theta = a * 0.01745329251;
  float cosTheta2 = (float)cos(theta) * 0.5f;
  float sinTheta2 = (float)sin(theta) * 0.5f;
  pts[0].x = xc + sinTheta2 * h + cosTheta2 * w;
  pts[0].y = yc + cosTheta2 * h - sinTheta2 * w;
  pts[1].x = xc - sinTheta2 * h + cosTheta2 * w;
  pts[1].y = yc - cosTheta2 * h - sinTheta2 * w;
  pts[2].x = 2 * xc - pts[0].x;
  pts[2].y = 2 * yc - pts[0].y;
  pts[3].x = 2 * xc - pts[1].x;
  pts[3].y = 2 * yc - pts[1].y;
}

static int orc_intersection_points(const orc_pt* pts1, const orc_pt* pts2, orc_pt* inter) {
  /* :79-164 */
  orc_pt vec1[4], vec2[4];
  for (int i = 0; i < 4; i++) {
    vec1[i] = orc_sub(pts1[(i + 1) % 4], pts1[i]);
    vec2[i] = orc_sub(pts2[(i + 1) % 4], pts2[i]);
  }
  double EPS = 1e-5;
  int num = 0;
  for (int i = 0; i < 4; i++) {
    for (int j = 0; j < 4; j++) {
      float det = orc_cross(vec2[j], vec1[i]);
      if (fabs(det) <= 1e-14) continue;
      orc_pt vec12 = orc_sub(pts2[j], pts1[i]);
      float t1 = orc_cross(vec2[j], vec12) / det;
      float t2 = orc_cross(vec1[i], vec12) / det;
      if (t1 > -EPS && t1 < 1.0f + EPS && t2 > -EPS && t2 < 1.0f + EPS) {
        inter[num].x = pts1[i].x + vec1[i].x * t1;
        inter[num].y = pts1[i].y + vec1[i].y * t1;
        num++;
      }
    }
  }
  {
    orc_pt AB = vec2[0], DA = vec2[3];
    float ABdotAB = orc_dot(AB, AB), ADdotAD = orc_dot(DA, DA);
    for (int i = 0; i < 4; i++) {
      orc_pt AP = orc_sub(pts1[i], pts2[0]);
      float APdotAB = orc_dot(AP, AB);
      float APdotAD = -orc_dot(AP, DA);
      if ((APdotAB > -EPS) && (APdotAD > -EPS) && (APdotAB < ABdotAB + EPS) &&
          (APdotAD < ADdotAD + EPS))
        inter[num++] = pts1[i];
    }
  }
  {
    orc_pt AB = vec1[0], DA = vec1[3];
    float ABdotAB = orc_dot(AB, AB), ADdotAD = orc_dot(DA, DA);
    for (int i = 0; i < 4; i++) {
      orc_pt AP = orc_sub(pts2[i], pts1[0]);
      float APdotAB = orc_dot(AP, AB);
      float APdotAD = -orc_dot(AP, DA);
      if ((APdotAB > -EPS) && (APdotAD > -EPS) && (APdotAB < ABdotAB + EPS) &&
          (APdotAD < ADdotAD + EPS))
        inter[num++] = pts2[i];
    }
  }
  return num;
}

static int orc_convex_hull(const orc_pt* p, int num_in, orc_pt* q) {
  /* :167-320, shift_to_zero = true (as called from :358) */
  int t = 0;
  for (int i = 1; i < num_in; i++)
    if (p[i].y < p[t].y || (p[i].y == p[t].y && p[i].x < p[t].x)) t = i;
  orc_pt start = p[t];
  for (int i = 0; i < num_in; i++) q[i] = orc_sub(p[i], start);
  orc_pt tmp = q[0]; q[0] = q[t]; q[t] = tmp;
  float dist[24];
  for (int i = 0; i < num_in; i++) dist[i] = orc_dot(q[i], q[i]);
  for (int i = 1; i < num_in - 1; i++)
    for (int j = i + 1; j < num_in; j++) {
      float cp = orc_cross(q[i], q[j]);
      if ((cp < -1e-6) || (fabs(cp) < 1e-6 && dist[i] > dist[j])) {
        orc_pt qt = q[i]; q[i] = q[j]; q[j] = qt;
        float dt = dist[i]; dist[i] = dist[j]; dist[j] = dt;
      }
    }
  for (int i = 0; i < num_in; i++) dist[i] = orc_dot(q[i], q[i]);
  int k;
  for (k = 1; k < num_in; k++)
    if (dist[k] > 1e-8) break;
  if (k == num_in) { q[0] = p[t]; return 1; }
  q[1] = q[k];
  int m = 2;
  for (int i = k + 1; i < num_in; i++) {
    while (m > 1) {
      orc_pt q1 = orc_sub(q[i], q[m - 2]), q2 = orc_sub(q[m - 1], q[m - 2]);
      if (q1.x * q2.y >= q2.x * q1.y) m--; else break;
    }
    q[m++] = q[i];
  }
  return m;
}

static float orc_polygon_area(const orc_pt* q, int m) {
  /* :323-334 */
  if (m <= 2) return 0;
  float area = 0;
  for (int i = 1; i < m - 1; i++)
    area += fabs(orc_cross(orc_sub(q[i], q[0]), orc_sub(q[i + 1], q[0])));
  return (float)(area / 2.0);
}

float orc_single_box_iou_rotated(const float* box1_raw, const float* box2_raw) {
  /* :363-390 */
  double center_shift_x = (box1_raw[0] + box2_raw[0]) / 2.0;
  double center_shift_y = (box1_raw[1] + box2_raw[1]) / 2.0;
  float x1 = (float)(box1_raw[0] - center_shift_x), y1 = (float)(box1_raw[1] - center_shift_y);
  float x2 = (float)(box2_raw[0] - center_shift_x), y2 = (float)(box2_raw[1] - center_shift_y);
  float w1 = box1_raw[2], h1 = box1_raw[3], a1 = box1_raw[4];
  float w2 = box2_raw[2], h2 = box2_raw[3], a2 = box2_raw[4];
  float area1 = w1 * h1, area2 = w2 * h2;
  if (area1 < 1e-14 || area2 < 1e-14) return 0.f;
  orc_pt pts1[4], pts2[4], ipts[24], ordered[24];
  orc_rot_vertices(x1, y1, w1, h1, a1, pts1);
  orc_rot_vertices(x2, y2, w2, h2, a2, pts2);
  int num = orc_intersection_points(pts1, pts2, ipts);
  float inter;
  if (num <= 2) {
    inter = 0.0f;
  } else {
    int nc = orc_convex_hull(ipts, num, ordered);
    inter = orc_polygon_area(ordered, nc);
  }
  return inter / (area1 + area2 - inter);
}

/* box_iou_rotated_cpu.cpp:8-37; always fp32 output */
int orc_box_iou_rotated(const float* b1, int n, const float* b2, int m, float* out) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++)
      out[(size_t)i * m + j] = orc_single_box_iou_rotated(b1 + 5 * i, b2 + 5 * j);
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Sorting helper: stable descending order of scores (ties: lower index first).  The
 * reference uses scores.sort(0, descending=true) (nms_rotated_cpu.cpp:26); tie order is not
 * pinned by its tests, benchmarks use distinct scores. */
typedef struct { float s; int64_t i; } orc_si;
static int orc_cmp_desc(const void* a, const void* b) {
  const orc_si* x = (const orc_si*)a; const orc_si* y = (const orc_si*)b;
  if (x->s > y->s) return -1;
  if (x->s < y->s) return 1;
  if (isnan(x->s) && !isnan(y->s)) return -1; /* torch sorts NaN as largest */
  if (!isnan(x->s) && isnan(y->s)) return 1;
  return (x->i < y->i) ? -1 : (x->i > y->i);
}
static int64_t* orc_argsort_desc(const float* scores, int64_t n) {
  orc_si* v = (orc_si*)malloc(sizeof(orc_si) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) { v[i].s = scores[i]; v[i].i = i; }
  qsort(v, (size_t)n, sizeof(orc_si), orc_cmp_desc);
  int64_t* order = (int64_t*)malloc(sizeof(int64_t) * (size_t)(n > 0 ? n : 1));
  for (int64_t i = 0; i < n; i++) order[i] = v[i].i;
  free(v);
  return order;
}

/* Rotated NMS.  nms_rotated_cpu.cpp:7-60: suppress when ovr >= iou_threshold (double compare).
 * Returns the number kept; keep[] holds original indices in decreasing score order. */
int64_t orc_nms_rotated(const float* dets, const float* scores, int64_t n, double iou_threshold,
                        int64_t* keep) {
  if (n == 0) return 0;
  int64_t* order = orc_argsort_desc(scores, n);
  uint8_t* suppressed = (uint8_t*)calloc((size_t)n, 1);
  int64_t num = 0;
  for (int64_t _i = 0; _i < n; _i++) {
    int64_t i = order[_i];
    if (suppressed[i]) continue;
    keep[num++] = i;
    for (int64_t _j = _i + 1; _j < n; _j++) {
      int64_t j = order[_j];
      if (suppressed[j]) continue;
      float ovr = orc_single_box_iou_rotated(dets + 5 * i, dets + 5 * j);
      if (ovr >= iou_threshold) suppressed[j] = 1;
    }
  }
  free(order); free(suppressed);
  return num;
}

/* Axis-aligned NMS.  Call site detectron2/layers/nms.py:6 -> torchvision.ops.nms; restates
 * torchvision's nms_kernel_impl (csrc/ops/cpu/nms_kernel.cpp): area=(x2-x1)*(y2-y1),
 * suppress when inter/(iarea+area_j-inter) > iou_threshold (double compare).  Same semantics
 * as the reference's own greedy oracle tests/layers/test_nms_rotated.py:44-66. */
int64_t orc_nms(const float* dets, const float* scores, int64_t n, double iou_threshold,
                int64_t* keep) {
  if (n == 0) return 0;
  int64_t* order = orc_argsort_desc(scores, n);
  uint8_t* suppressed = (uint8_t*)calloc((size_t)n, 1);
  float* areas = (float*)malloc(sizeof(float) * (size_t)n);
  for (int64_t i = 0; i < n; i++)
    areas[i] = (dets[4 * i + 2] - dets[4 * i]) * (dets[4 * i + 3] - dets[4 * i + 1]);
  int64_t num = 0;
  for (int64_t _i = 0; _i < n; _i++) {
    int64_t i = order[_i];
    if (suppressed[i]) continue;
    keep[num++] = i;
    float ix1 = dets[4 * i], iy1 = dets[4 * i + 1], ix2 = dets[4 * i + 2], iy2 = dets[4 * i + 3];
    float iarea = areas[i];
    for (int64_t _j = _i + 1; _j < n; _j++) {
      int64_t j = order[_j];
      if (suppressed[j]) continue;
      /* std::max(a,b) = (a<b)?b:a ; std::min(a,b) = (b<a)?b:a  (NaN behaviour kept) */
      float xx1 = (ix1 < dets[4 * j]) ? dets[4 * j] : ix1;             /* std::max(ix1, x1[j]) */
      float yy1 = (iy1 < dets[4 * j + 1]) ? dets[4 * j + 1] : iy1;
      float xx2 = (dets[4 * j + 2] < ix2) ? dets[4 * j + 2] : ix2;     /* std::min(ix2, x2[j]) */
      float yy2 = (dets[4 * j + 3] < iy2) ? dets[4 * j + 3] : iy2;
      float w = (0.f < (xx2 - xx1)) ? (xx2 - xx1) : 0.f; /* std::max(0, xx2 - xx1) */
      float h = (0.f < (yy2 - yy1)) ? (yy2 - yy1) : 0.f;
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (ovr > iou_threshold) suppressed[j] = 1;
    }
  }
  free(order); free(suppressed); free(areas);
  return num;
}

/* batched_nms.  detectron2/layers/nms.py:11-22 -> torchvision.ops.boxes.batched_nms on
 * boxes.float().  Restated as its `_batched_nms_vanilla` strategy: NMS independently per
 * category on the ORIGINAL coordinates, union of kept, sorted by decreasing score (the
 * coordinate-offset strategy differs only by fp rounding of the shifted coordinates).
 * rotated != 0 gives batched_nms_rotated semantics per category (nms.py:96-147 uses the
 * offset trick; same per-category semantics up to that rounding). */
int64_t orc_batched_nms(const float* dets, const float* scores, const int64_t* idxs, int64_t n,
                        double iou_threshold, int rotated, int64_t* keep) {
  if (n == 0) return 0;
  int bw = rotated ? 5 : 4;
  uint8_t* keep_mask = (uint8_t*)calloc((size_t)n, 1);
  uint8_t* done = (uint8_t*)calloc((size_t)n, 1);
  float* cd = (float*)malloc(sizeof(float) * (size_t)n * bw);
  float* cs = (float*)malloc(sizeof(float) * (size_t)n);
  int64_t* cmap = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  int64_t* ck = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  for (int64_t s = 0; s < n; s++) {
    if (done[s]) continue;
    int64_t cls = idxs[s], cn = 0;
    for (int64_t j = s; j < n; j++)
      if (!done[j] && idxs[j] == cls) {
        done[j] = 1;
        memcpy(cd + cn * bw, dets + j * bw, sizeof(float) * bw);
        cs[cn] = scores[j];
        cmap[cn++] = j;
      }
    int64_t nk = rotated ? orc_nms_rotated(cd, cs, cn, iou_threshold, ck)
                         : orc_nms(cd, cs, cn, iou_threshold, ck);
    for (int64_t t = 0; t < nk; t++) keep_mask[cmap[ck[t]]] = 1;
  }
  int64_t* order = orc_argsort_desc(scores, n);
  int64_t num = 0;
  for (int64_t r = 0; r < n; r++)
    if (keep_mask[order[r]]) keep[num++] = order[r];
  free(order); free(keep_mask); free(done); free(cd); free(cs); free(cmap); free(ck);
  return num;
}

/* ------------------------------------------------------------------------------------ */
/* paste_masks_in_image.  detectron2/layers/mask_ops.py:17-69 (_do_paste_mask) and :74-147,
 * CPU path: one mask per chunk, skip_empty=True (bbox region only, zeros elsewhere).
 * F.grid_sample(bilinear, zeros padding, align_corners=False) is restated as ATen's CPU kernel
 * evaluates it on an FMA-capable x86 build (aten/src/ATen/native/cpu/GridSamplerKernel.cpp,
 * not part of the reference tree; torch 2.10 in this image) -- determined empirically against
 * the reference python run here (0 differing bits in the pre-threshold value):
 *   x  = fma(gx + 1, W/2, -0.5)                    (unnormalize, align_corners=False)
 *   w  = x - floor(x); e = 1 - w; n = y - floor(y); s = 1 - n
 *   v  = fma(se_val, n*w, fma(sw_val, n*e, fma(ne_val, s*w, nw_val * (s*e))))
 * out-of-range corners contribute 0.  The grid itself is built by separate torch ops
 * (mask_ops.py:51-54), i.e. without contraction.
 * out: uint8 (N, img_h, img_w); threshold >= 0 -> 0/1 (bool), else trunc(value*255). */
static float orc_mask_at(const float* m, int mh, int mw, int y, int x) {
  return (y >= 0 && y < mh && x >= 0 && x < mw) ? m[y * mw + x] : 0.f;
}
float orc_paste_sample(const float* mask, int mh, int mw, float x0, float y0, float x1, float y1,
                       int px, int py) {
  float img_y = (float)py + 0.5f;
  float img_x = (float)px + 0.5f;
  float gy = (img_y - y0) / (y1 - y0) * 2.f - 1.f;
  float gx = (img_x - x0) / (x1 - x0) * 2.f - 1.f;
  float ix = fmaf(gx + 1.f, (float)mw / 2.f, -0.5f);
  float iy = fmaf(gy + 1.f, (float)mh / 2.f, -0.5f);
  float fx = floorf(ix), fy = floorf(iy);
  /* coordinates that do not fit an int (inf/NaN/huge) sample nothing */
  if (!(fx > -4.0e8f && fx < 4.0e8f && fy > -4.0e8f && fy < 4.0e8f)) return 0.f;
  int x_w = (int)fx, y_n = (int)fy;
  float w = ix - fx, e = 1.f - w;
  float n = iy - fy, s = 1.f - n;
  float nw = s * e, ne = s * w, sw = n * e, se = n * w;
  float v = orc_mask_at(mask, mh, mw, y_n, x_w) * nw;
  v = fmaf(orc_mask_at(mask, mh, mw, y_n, x_w + 1), ne, v);
  v = fmaf(orc_mask_at(mask, mh, mw, y_n + 1, x_w), sw, v);
  v = fmaf(orc_mask_at(mask, mh, mw, y_n + 1, x_w + 1), se, v);
  return v;
}

static int orc_clampi(float v, int lo, int hi, int use_lo) {
  /* torch.clamp(...).to(int32) of an already floor()/ceil()-ed float */
  if (use_lo) { if (v < (float)lo) v = (float)lo; } else { if (v > (float)hi) v = (float)hi; }
  return (int)v;
}

int orc_paste_masks_ex(const float* masks, const float* boxes, int n, int mh, int mw, int img_h,
                       int img_w, float threshold, uint8_t* out, float* soft_out /* may be NULL */,
                       int skip_empty);
int orc_paste_masks(const float* masks, const float* boxes, int n, int mh, int mw, int img_h,
                    int img_w, float threshold, uint8_t* out, float* soft_out /* may be NULL */) {
  return orc_paste_masks_ex(masks, boxes, n, mh, mw, img_h, img_w, threshold, out, soft_out, 1);
}
/* skip_empty = 1: the CPU path (one mask per chunk, bbox region only, mask_ops.py:116-119,134);
 * skip_empty = 0: the device path of the reference -- the whole image is sampled. */
int orc_paste_masks_ex(const float* masks, const float* boxes, int n, int mh, int mw, int img_h,
                       int img_w, float threshold, uint8_t* out, float* soft_out /* may be NULL */,
                       int skip_empty) {
  memset(out, 0, (size_t)n * img_h * img_w);
  if (soft_out) memset(soft_out, 0, sizeof(float) * (size_t)n * img_h * img_w);
  for (int k = 0; k < n; k++) {
    const float* b = boxes + 4 * k;
    const float* m = masks + (size_t)k * mh * mw;
    /* mask_ops.py:38-43 with a single box */
    int x0i = orc_clampi(floorf(b[0]) - 1.f, 0, 0, 1);
    int y0i = orc_clampi(floorf(b[1]) - 1.f, 0, 0, 1);
    int x1i = orc_clampi(ceilf(b[2]) + 1.f, 0, img_w, 0);
    int y1i = orc_clampi(ceilf(b[3]) + 1.f, 0, img_h, 0);
    if (!skip_empty) { x0i = 0; y0i = 0; x1i = img_w; y1i = img_h; }
    for (int py = y0i; py < y1i; py++)
      for (int px = x0i; px < x1i; px++) {
        float v = orc_paste_sample(m, mh, mw, b[0], b[1], b[2], b[3], px, py);
        size_t o = ((size_t)k * img_h + py) * img_w + px;
        if (soft_out) soft_out[o] = v;
        if (threshold >= 0)
          out[o] = (v >= threshold) ? 1 : 0;
        else
          out[o] = (uint8_t)(v * 255.f);
      }
  }
  return 0;
}

/* ------------------------------------------------------------------------------------ */
/* Deformable convolution v1 / v2 (modulated).  The reference has NO CPU implementation of
 * v2 (layers/deform_conv.py:210-211; csrc/deformable/deform_conv.h:311,374) and only a
 * torchvision fallback for v1 forward.  This restates the CUDA kernels
 * csrc/deformable/deform_conv_cuda_kernel.cu:96-130 (bilinear), :216-288 / :785-868 (im2col),
 * :291-363 / :870-949 (col2im), :366-452 / :951-1066 (coord + mask grads) and the host GEMM
 * flow csrc/deformable/deform_conv_cuda.cu:272-824 (v1) / :826-1221 (v2).
 * GEMMs accumulate in double and round once to fp32 (the reference uses at::addmm_).
 * mask == NULL -> v1 (no modulation).  Offsets: channel 2k = dh, 2k+1 = dw for tap
 * k = i*kw + j (deform_conv_cuda_kernel.cu:263-269). */
typedef struct {
  int B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, Ho, Wo;
} orc_dcn_shape;

static float orc_dcn_bilinear(const float* im, int data_width, int height, int width, float h,
                              float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high = h_low + 1, w_high = w_low + 1;
  float lh = h - h_low, lw = w - w_low;
  float hh = 1 - lh, hw = 1 - lw;
  float v1 = 0, v2 = 0, v3 = 0, v4 = 0;
  if (h_low >= 0 && w_low >= 0) v1 = im[h_low * data_width + w_low];
  if (h_low >= 0 && w_high <= width - 1) v2 = im[h_low * data_width + w_high];
  if (h_high <= height - 1 && w_low >= 0) v3 = im[h_high * data_width + w_low];
  if (h_high <= height - 1 && w_high <= width - 1) v4 = im[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
}

/* columns[(c*kh*kw + tap), b, ho, wo] for one image b (batch dim dropped): size C*kh*kw*Ho*Wo */
static void orc_dcn_im2col(const orc_dcn_shape* s, const float* x_b, const float* off_b,
                           const float* mask_b, float* col) {
  int L = s->Ho * s->Wo, K2 = s->kh * s->kw, cpg = s->C / s->dg;
  for (int c = 0; c < s->C; c++) {
    int g = c / cpg;
    const float* im = x_b + (size_t)c * s->H * s->W;
    const float* off = off_b + (size_t)g * 2 * K2 * L;
    const float* msk = mask_b ? mask_b + (size_t)g * K2 * L : NULL;
    for (int i = 0; i < s->kh; i++)
      for (int j = 0; j < s->kw; j++) {
        int tap = i * s->kw + j;
        float* dst = col + ((size_t)c * K2 + tap) * L;
        for (int ho = 0; ho < s->Ho; ho++)
          for (int wo = 0; wo < s->Wo; wo++) {
            int l = ho * s->Wo + wo;
            float oh = off[(size_t)(2 * tap) * L + l], ow = off[(size_t)(2 * tap + 1) * L + l];
            float h_im = (ho * s->sh - s->ph) + i * s->dh + oh;
            float w_im = (wo * s->sw - s->pw) + j * s->dw + ow;
            float val = 0.f;
            if (h_im > -1 && w_im > -1 && h_im < s->H && w_im < s->W)
              val = orc_dcn_bilinear(im, s->W, s->H, s->W, h_im, w_im);
            dst[l] = msk ? val * msk[(size_t)tap * L + l] : val;
          }
      }
  }
}

int orc_deform_conv_forward(const float* x, const float* offset, const float* mask,
                            const float* weight, const float* bias, int B, int C, int H, int W,
                            int Co, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                            int groups, int dg, float* out) {
  orc_dcn_shape s = {B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, 0, 0};
  s.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  s.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  int L = s.Ho * s.Wo, K2 = kh * kw;
  int Cg = C / groups, Cog = Co / groups, Kg = Cg * K2;
  float* col = (float*)malloc(sizeof(float) * (size_t)C * K2 * L);
  for (int b = 0; b < B; b++) {
    orc_dcn_im2col(&s, x + (size_t)b * C * H * W, offset + (size_t)b * dg * 2 * K2 * L,
                   mask ? mask + (size_t)b * dg * K2 * L : NULL, col);
    for (int g = 0; g < groups; g++)
      for (int co = 0; co < Cog; co++) {
        const float* wrow = weight + ((size_t)(g * Cog + co)) * Kg;
        float* o = out + (((size_t)b * Co + g * Cog + co)) * L;
        for (int l = 0; l < L; l++) {
          double acc = 0.0;
          for (int k = 0; k < Kg; k++) acc += (double)wrow[k] * col[((size_t)g * Kg + k) * L + l];
          float r = (float)acc;
          if (bias) r = r + bias[g * Cog + co];
          o[l] = r;
        }
      }
  }
  free(col);
  return 0;
}

static float orc_dcn_grad_weight(float ah, float aw, int h, int w, int height, int width) {
  /* deform_conv_cuda_kernel.cu:132-162 */
  if (ah <= -1 || ah >= height || aw <= -1 || aw >= width) return 0;
  int hl = (int)floorf(ah), wl = (int)floorf(aw);
  int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (h == hl && w == wl) weight = (h + 1 - ah) * (w + 1 - aw);
  if (h == hl && w == wh) weight = (h + 1 - ah) * (aw + 1 - w);
  if (h == hh && w == wl) weight = (ah + 1 - h) * (w + 1 - aw);
  if (h == hh && w == wh) weight = (ah + 1 - h) * (aw + 1 - w);
  return weight;
}

static float orc_dcn_coord_weight(float ah, float aw, int height, int width, const float* im,
                                  int data_width, int bp_dir) {
  /* deform_conv_cuda_kernel.cu:164-214 */
  if (ah <= -1 || ah >= height || aw <= -1 || aw >= width) return 0;
  int hl = (int)floorf(ah), wl = (int)floorf(aw);
  int hh = hl + 1, wh = wl + 1;
  float weight = 0;
  if (bp_dir == 0) {
    if (hl >= 0 && wl >= 0) weight += -1 * (wl + 1 - aw) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += -1 * (aw - wl) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += (wl + 1 - aw) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (aw - wl) * im[hh * data_width + wh];
  } else {
    if (hl >= 0 && wl >= 0) weight += -1 * (hl + 1 - ah) * im[hl * data_width + wl];
    if (hl >= 0 && wh <= width - 1) weight += (hl + 1 - ah) * im[hl * data_width + wh];
    if (hh <= height - 1 && wl >= 0) weight += -1 * (ah - hl) * im[hh * data_width + wl];
    if (hh <= height - 1 && wh <= width - 1) weight += (ah - hl) * im[hh * data_width + wh];
  }
  return weight;
}

/* All grad_* buffers must be zero-filled by the caller (deform_conv.py:97-98,121,250-254).
 * Any of grad_input/grad_offset/grad_mask/grad_weight/grad_bias may be NULL to skip. */
int orc_deform_conv_backward(const float* x, const float* offset, const float* mask,
                             const float* weight, const float* grad_out, int B, int C, int H,
                             int W, int Co, int kh, int kw, int sh, int sw, int ph, int pw,
                             int dh, int dw, int groups, int dg, float* grad_input,
                             float* grad_offset, float* grad_mask, float* grad_weight,
                             float* grad_bias) {
  orc_dcn_shape s = {B, C, H, W, Co, kh, kw, sh, sw, ph, pw, dh, dw, groups, dg, 0, 0};
  s.Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1;
  s.Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;
  int L = s.Ho * s.Wo, K2 = kh * kw;
  int Cg = C / groups, Cog = Co / groups, Kg = Cg * K2, cpg = C / dg;
  float* col = (float*)malloc(sizeof(float) * (size_t)C * K2 * L);
  double* gw_acc = grad_weight ? (double*)calloc((size_t)Co * Kg, sizeof(double)) : NULL;
  double* gb_acc = grad_bias ? (double*)calloc((size_t)Co, sizeof(double)) : NULL;
  for (int b = 0; b < B; b++) {
    const float* x_b = x + (size_t)b * C * H * W;
    const float* off_b = offset + (size_t)b * dg * 2 * K2 * L;
    const float* mask_b = mask ? mask + (size_t)b * dg * K2 * L : NULL;
    const float* go_b = grad_out + (size_t)b * Co * L;
    /* columns = W^T * grad_out  (deform_conv_cuda.cu:1097-1103) */
    for (int g = 0; g < groups; g++)
      for (int k = 0; k < Kg; k++)
        for (int l = 0; l < L; l++) {
          double acc = 0.0;
          for (int co = 0; co < Cog; co++)
            acc += (double)weight[((size_t)(g * Cog + co)) * Kg + k] *
                go_b[((size_t)(g * Cog + co)) * L + l];
          col[((size_t)g * Kg + k) * L + l] = (float)acc;
        }
    /* coord (+mask) gradient: deform_conv_cuda_kernel.cu:366-452 / :951-1066 */
    if (grad_offset || grad_mask) {
      for (int g = 0; g < dg; g++)
        for (int tap = 0; tap < K2; tap++) {
          int i = tap / kw, j = tap % kw;
          for (int ho = 0; ho < s.Ho; ho++)
            for (int wo = 0; wo < s.Wo; wo++) {
              int l = ho * s.Wo + wo;
              float oh = off_b[((size_t)g * 2 * K2 + 2 * tap) * L + l];
              float ow = off_b[((size_t)g * 2 * K2 + 2 * tap + 1) * L + l];
              float m = mask_b ? mask_b[((size_t)g * K2 + tap) * L + l] : 1.f;
              float inv_h = (ho * sh - ph) + i * dh + oh;
              float inv_w = (wo * sw - pw) + j * dw + ow;
              int inside = !(inv_h <= -1 || inv_w <= -1 || inv_h >= H || inv_w >= W);
              float vh = 0, vw = 0, mval = 0;
              for (int cc = 0; cc < cpg; cc++) {
                int c = g * cpg + cc;
                const float* im = x_b + (size_t)c * H * W;
                float dc = col[((size_t)c * K2 + tap) * L + l];
                float ih = inside ? inv_h : -2.f, iw = inside ? inv_w : -2.f;
                if (inside) mval += dc * orc_dcn_bilinear(im, W, H, W, inv_h, inv_w);
                float wgt_h = orc_dcn_coord_weight(ih, iw, H, W, im, W, 0);
                float wgt_w = orc_dcn_coord_weight(ih, iw, H, W, im, W, 1);
                if (mask_b) { vh += wgt_h * dc * m; vw += wgt_w * dc * m; }
                else { vh += wgt_h * dc; vw += wgt_w * dc; }
              }
              if (grad_offset) {
                float* go = grad_offset + (size_t)b * dg * 2 * K2 * L;
                go[((size_t)g * 2 * K2 + 2 * tap) * L + l] = vh;
                go[((size_t)g * 2 * K2 + 2 * tap + 1) * L + l] = vw;
              }
              if (grad_mask && mask_b)
                grad_mask[(size_t)b * dg * K2 * L + ((size_t)g * K2 + tap) * L + l] = mval;
            }
        }
    }
    /* col2im: deform_conv_cuda_kernel.cu:291-363 / :870-949 */
    if (grad_input) {
      float* gi_b = grad_input + (size_t)b * C * H * W;
      for (int c = 0; c < C; c++) {
        int g = c / cpg;
        for (int tap = 0; tap < K2; tap++) {
          int i = tap / kw, j = tap % kw;
          for (int ho = 0; ho < s.Ho; ho++)
            for (int wo = 0; wo < s.Wo; wo++) {
              int l = ho * s.Wo + wo;
              float oh = off_b[((size_t)g * 2 * K2 + 2 * tap) * L + l];
              float ow = off_b[((size_t)g * 2 * K2 + 2 * tap + 1) * L + l];
              float m = mask_b ? mask_b[((size_t)g * K2 + tap) * L + l] : 1.f;
              float ch = (ho * sh - ph) + i * dh + oh;
              float cw = (wo * sw - pw) + j * dw + ow;
              float top = col[((size_t)c * K2 + tap) * L + l];
              if (mask_b) top = top * m;
              int cur_h = (int)ch, cur_w = (int)cw;
              for (int dy = -2; dy <= 2; dy++)
                for (int dx = -2; dx <= 2; dx++)
                  if (cur_h + dy >= 0 && cur_h + dy < H && cur_w + dx >= 0 && cur_w + dx < W &&
                      fabsf(ch - (cur_h + dy)) < 1 && fabsf(cw - (cur_w + dx)) < 1) {
                    float wgt = orc_dcn_grad_weight(ch, cw, cur_h + dy, cur_w + dx, H, W);
                    gi_b[((size_t)c * H + cur_h + dy) * W + cur_w + dx] += wgt * top;
                  }
            }
        }
      }
    }
    /* dW += grad_out * col^T with the re-gathered (modulated) columns; dbias += sum grad_out */
    if (grad_weight || grad_bias) {
      if (grad_weight) orc_dcn_im2col(&s, x_b, off_b, mask_b, col);
      for (int g = 0; g < groups; g++)
        for (int co = 0; co < Cog; co++) {
          const float* gor = go_b + ((size_t)(g * Cog + co)) * L;
          if (grad_weight)
            for (int k = 0; k < Kg; k++) {
              const float* cr = col + ((size_t)g * Kg + k) * L;
              double acc = 0.0;
              for (int l = 0; l < L; l++) acc += (double)gor[l] * cr[l];
              gw_acc[((size_t)(g * Cog + co)) * Kg + k] += acc;
            }
          if (grad_bias) {
            double acc = 0.0;
            for (int l = 0; l < L; l++) acc += gor[l];
            gb_acc[g * Cog + co] += acc;
          }
        }
    }
  }
  if (grad_weight) for (size_t t = 0; t < (size_t)Co * Kg; t++) grad_weight[t] += (float)gw_acc[t];
  if (grad_bias) for (int t = 0; t < Co; t++) grad_bias[t] += (float)gb_acc[t];
  free(col); free(gw_acc); free(gb_acc);
  return 0;
}

#ifdef __cplusplus
}
#endif

/* ---------------------------------------------------------------------------------------------
 * Matcher.__call__ + set_low_quality_matches_   (detectron2/modeling/matcher.py:62-127)
 * q [M,N] row-major match-quality matrix; thr [T] (the constructor's thresholds, without the
 * -inf / +inf sentinels); lab [T+1]; matches [N] int64; out [N] int8.
 *   matcher.py:80-90   M == 0 (numel == 0): matches = 0, labels = lab[0]
 *   matcher.py:94      matched_vals, matches = q.max(dim=0)  -- first maximal index, NaN is maximal
 *   matcher.py:96-101  labels start at 1; every interval low <= v < high overwrites with its label
 *   matcher.py:103-127 low-quality: every (g, n) with q[g][n] == max_n q[g][:] -> label 1
 * Parity: pinned against the reference class itself (tests/golden/matcher.npz, generated by
 * tests/golden/make_golden.py from /root/reference/detectron2/modeling/matcher.py). */
static int orc_isnan(float v) { return v != v; }
void orc_matcher(const float* q, int M, int N, const float* thr, const signed char* lab, int T, int allow_low,
                 long long* matches, signed char* out) {
  if (M == 0 || N == 0) {
    for (int n = 0; n < N; n++) { matches[n] = 0; out[n] = lab[0]; }
    return;
  }
  for (int n = 0; n < N; n++) {
    float best = q[n];
    int bi = 0;
    for (int g = 1; g < M; g++) {
      const float v = q[(long)g * N + n];
      if ((v > best) || (orc_isnan(v) && !orc_isnan(best))) { best = v; bi = g; }
    }
    signed char l = 1;
    for (int k = 0; k <= T; k++) {
      const int ge_low = (k == 0) ? (best >= -INFINITY) : (best >= thr[k - 1]);
      const int lt_high = (k == T) ? (best < INFINITY) : (best < thr[k]);
      if (ge_low && lt_high) l = lab[k];
    }
    matches[n] = bi;
    out[n] = l;
  }
  if (allow_low) {
    for (int g = 0; g < M; g++) {
      float hi = q[(long)g * N];
      for (int n = 1; n < N; n++) {
        const float v = q[(long)g * N + n];
        if ((v > hi) || (orc_isnan(v) && !orc_isnan(hi))) hi = v;
      }
      for (int n = 0; n < N; n++)
        if (q[(long)g * N + n] == hi) out[n] = 1;
    }
  }
}

/* ------------------------------------------------------------------------------------ */
/* Polygon rasterisation: detectron2/structures/masks.py:20-36 (polygons_to_bitmask) calls pycocotools
 * `mask_util.frPyObjects` + `merge` + `decode`.  pycocotools is a third-party dependency that is NOT in
 * /root/reference (setup.py install_requires "pycocotools>=2.0.2") and not installed here: PARITY UNPINNED.
 * This restates the published algorithm of cocoapi `common/maskApi.c` (pycocotools 2.0.x): rleFrPoly -- upsample by
 * 5, walk every edge densely with the rounded-slope stepping, keep the points where the x coordinate changes as
 * y-boundary crossings, downsample, sort the column-major positions x*h + y, take differences as run lengths
 * (zero-length runs merge) -- then rleMerge (union) and rleDecode (column-major runs starting with 0).
 * A pixel with column-major index t is therefore set iff an ODD number of crossings has position <= t; the union of
 * several polygons is the OR of their masks.  Anchored on the reference's own known answer
 * tests/structures/test_masks.py:31-38 (an integer box polygon fills exactly [x0, x1) x [y0, y1)).
 * out: uint8 (h, w) row-major, OR-ed into (the caller zeroes it); xy: k vertices (x, y) in double. */
static int orc_int_of(double v) { /* (int) of a double as x86 does it: NaN / out of range -> INT_MIN */
  if (!(v > -2147483649.0 && v < 2147483648.0)) return (int)0x80000000;
  return (int)v;
}
int orc_poly_to_mask(const double* xy, int k, int h, int w, uint8_t* out) {
  if (k <= 0 || h <= 0 || w <= 0) return 0;
  const double scale = 5;
  int* x = (int*)malloc(sizeof(int) * (size_t)(k + 1));
  int* y = (int*)malloc(sizeof(int) * (size_t)(k + 1));
  for (int j = 0; j < k; j++) x[j] = orc_int_of(scale * xy[j * 2 + 0] + .5);
  x[k] = x[0];
  for (int j = 0; j < k; j++) y[j] = orc_int_of(scale * xy[j * 2 + 1] + .5);
  y[k] = y[0];
  size_t m = 0;
  for (int j = 0; j < k; j++) {
    long ax = labs((long)x[j] - x[j + 1]), ay = labs((long)y[j] - y[j + 1]);
    m += (size_t)(ax > ay ? ax : ay) + 1;
  }
  int* u = (int*)malloc(sizeof(int) * m);
  int* v = (int*)malloc(sizeof(int) * m);
  m = 0;
  for (int j = 0; j < k; j++) {
    int xs = x[j], xe = x[j + 1], ys = y[j], ye = y[j + 1], dx, dy, t, d, flip;
    double s;
    dx = abs(xe - xs); dy = abs(ys - ye);
    flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
    if (flip) { t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
    s = dx >= dy ? (double)(ye - ys) / dx : (double)(xe - xs) / dy;
    if (dx >= dy) for (d = 0; d <= dx; d++) {
      t = flip ? dx - d : d; u[m] = t + xs; v[m] = orc_int_of(ys + s * t + .5); m++;
    } else for (d = 0; d <= dy; d++) {
      t = flip ? dy - d : d; v[m] = t + ys; u[m] = orc_int_of(xs + s * t + .5); m++;
    }
  }
  /* y-boundary crossings, downsampled; toggle counts per column-major position */
  unsigned* cnt = (unsigned*)calloc((size_t)h * w + 1, sizeof(unsigned));
  for (size_t j = 1; j < m; j++) if (u[j] != u[j - 1]) {
    double xd = (double)(u[j] < u[j - 1] ? u[j] : u[j] - 1); xd = (xd + .5) / scale - .5;
    if (floor(xd) != xd || xd < 0 || xd > w - 1) continue;
    double yd = (double)(v[j] < v[j - 1] ? v[j] : v[j - 1]); yd = (yd + .5) / scale - .5;
    if (yd < 0) yd = 0; else if (yd > h) yd = h; yd = ceil(yd);
    cnt[(size_t)((int)xd) * h + (int)yd]++;
  }
  unsigned par = 0;
  for (int t = 0; t < h * w; t++) {  /* column-major: t = col * h + row */
    par ^= cnt[t] & 1u;
    if (par) out[(size_t)(t % h) * w + (t / h)] = 1;
  }
  free(x); free(y); free(u); free(v); free(cnt);
  return 0;
}
