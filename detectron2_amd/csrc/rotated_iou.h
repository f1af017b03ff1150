// Rotated-box IoU on the device: polygon clip (<= 24 candidate points), Graham scan, shoelace.
// Same algorithm and the same float/double promotion points as the reference's shared
// host/device core detectron2/layers/csrc/box_iou_rotated/box_iou_rotated_utils.h:59-390, so
// that results are bit-identical to the CPU op.  Compiled with FP contraction OFF.
// The per-thread point lists live in LDS (column `tid` of [24][BLOCK] arrays): dynamic indexing
// into registers would go to scratch memory; the index-major layout is bank-conflict free.
#pragma once
#include "common.h"

namespace d2amd {

struct Pt { float x, y; };

__device__ __forceinline__ float cross2(Pt a, Pt b) { return a.x * b.y - b.x * a.y; }
__device__ __forceinline__ float dot2(Pt a, Pt b) { return a.x * b.x + a.y * b.y; }
__device__ __forceinline__ Pt sub2(Pt a, Pt b) { return Pt{a.x - b.x, a.y - b.y}; }

// r04: ONE point list per thread (the candidates are shifted in place into the hull work list, the squared distances
// are recomputed from the points -- the same expression on the same values: bit-identical): 12 KB per 64 threads
// instead of 30 KB, which was what bounded the occupancy of every rotated kernel (5 waves per CU).
template <int BLOCK>
struct RotIouScratch {
  float qx[24][BLOCK], qy[24][BLOCK];   // intersection candidates, then (shifted in place) the hull work list
};

__device__ __forceinline__ void rot_vertices(float xc, float yc, float w, float h, float a, Pt (&pts)[4]) {
  // utils.h:59-76 -- theta in double, cos/sin cast to float
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

// Quick reject, EXACT: two boxes whose centres are farther apart than the sum of their half diagonals (plus the
// reach of the reference's tolerances) produce no candidate point in single_box_iou_rotated, i.e. inter = 0 and the
// function returns 0.f / (area1 + area2) = +0.f -- returned without the clip.  The reach: an edge / edge parameter may
// exceed the segment by EPS = 1e-5 of its length, a vertex passes the "inside" test up to EPS / |side| outside the other
// box -- <= 1e-3 once every side is >= 0.01 (smaller boxes take the full path: for a 1e-7-sized box the reference's
// absolute EPS makes far-away points "inside", and its result must be reproduced, not corrected); vertex coordinates are
// rounded to ~1e-7 of the centre distance.  RRPN matching (16 x 268,569) rejects > 99 % of its pairs here.
// (x1, y1), (x2, y2): the centres relative to their midpoint, as single_box_iou_rotated computes them.
__device__ __forceinline__ bool rot_quick_reject(float x1, float y1, float w1, float h1, float x2, float y2, float w2,
                                                 float h2) {
  if (w1 >= 0.01f && h1 >= 0.01f && w2 >= 0.01f && h2 >= 0.01f) {
    const float r12 = 0.5f * (sqrtf(w1 * w1 + h1 * h1) + sqrtf(w2 * w2 + h2 * h2));
    const float R = r12 * 1.001f + 0.01f;
    const float ddx = x2 - x1, ddy = y2 - y1;
    if (ddx * ddx + ddy * ddy > R * R) return true;
  }
  return false;
}
// the same test from the two box records (what a caller that wants to skip the call altogether evaluates): true only
// where single_box_iou_rotated(b1, b2) returns exactly +0.f
__device__ __forceinline__ bool rot_pair_is_zero(const float* __restrict__ b1, const float* __restrict__ b2) {
  const double csx = (b1[0] + b2[0]) / 2.0;
  const double csy = (b1[1] + b2[1]) / 2.0;
  const float x1 = (float)(b1[0] - csx), y1 = (float)(b1[1] - csy);
  const float x2 = (float)(b2[0] - csx), y2 = (float)(b2[1] - csy);
  const float w1 = b1[2], h1 = b1[3], w2 = b2[2], h2 = b2[3];
  const float area1 = w1 * h1, area2 = w2 * h2;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return true;
  return rot_quick_reject(x1, y1, w1, h1, x2, y2, w2, h2);
}

template <int BLOCK>
__device__ float single_box_iou_rotated(const float* __restrict__ b1, const float* __restrict__ b2,
                                        RotIouScratch<BLOCK>& S, int tid) {
  // utils.h:363-390
  double csx = (b1[0] + b2[0]) / 2.0;
  double csy = (b1[1] + b2[1]) / 2.0;
  float x1 = (float)(b1[0] - csx), y1 = (float)(b1[1] - csy);
  float x2 = (float)(b2[0] - csx), y2 = (float)(b2[1] - csy);
  float w1 = b1[2], h1 = b1[3], a1 = b1[4];
  float w2 = b2[2], h2 = b2[3], a2 = b2[4];
  float area1 = w1 * h1, area2 = w2 * h2;
  if ((double)area1 < 1e-14 || (double)area2 < 1e-14) return 0.f;
  if (rot_quick_reject(x1, y1, w1, h1, x2, y2, w2, h2)) return 0.f;

  Pt pts1[4], pts2[4], vec1[4], vec2[4];
  rot_vertices(x1, y1, w1, h1, a1, pts1);
  rot_vertices(x2, y2, w2, h2, a2, pts2);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    vec1[i] = sub2(pts1[(i + 1) & 3], pts1[i]);
    vec2[i] = sub2(pts2[(i + 1) & 3], pts2[i]);
  }
  const double EPS = 1e-5;
  int num = 0;
  // utils.h:98-122 edge/edge intersections
#pragma unroll
  for (int i = 0; i < 4; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float det = cross2(vec2[j], vec1[i]);
      if (fabs((double)det) <= 1e-14) continue;
      Pt vec12 = sub2(pts2[j], pts1[i]);
      float t1 = cross2(vec2[j], vec12) / det;
      float t2 = cross2(vec1[i], vec12) / det;
      if ((double)t1 > -EPS && (double)t1 < (double)1.0f + EPS && (double)t2 > -EPS &&
          (double)t2 < (double)1.0f + EPS) {
        S.qx[num][tid] = pts1[i].x + vec1[i].x * t1;
        S.qy[num][tid] = pts1[i].y + vec1[i].y * t1;
        num++;
      }
    }
  }
  // utils.h:124-143 vertices of rect1 inside rect2
  {
    Pt AB = vec2[0], DA = vec2[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Pt AP = sub2(pts1[i], pts2[0]);
      float APdotAB = dot2(AP, AB);
      float APdotAD = -dot2(AP, DA);
      if (((double)APdotAB > -EPS) && ((double)APdotAD > -EPS) &&
          ((double)APdotAB < (double)ABdotAB + EPS) && ((double)APdotAD < (double)ADdotAD + EPS)) {
        S.qx[num][tid] = pts1[i].x;
        S.qy[num][tid] = pts1[i].y;
        num++;
      }
    }
  }
  // utils.h:145-161 vertices of rect2 inside rect1
  {
    Pt AB = vec1[0], DA = vec1[3];
    float ABdotAB = dot2(AB, AB), ADdotAD = dot2(DA, DA);
#pragma unroll
    for (int i = 0; i < 4; i++) {
      Pt AP = sub2(pts2[i], pts1[0]);
      float APdotAB = dot2(AP, AB);
      float APdotAD = -dot2(AP, DA);
      if (((double)APdotAB > -EPS) && ((double)APdotAD > -EPS) &&
          ((double)APdotAB < (double)ABdotAB + EPS) && ((double)APdotAD < (double)ADdotAD + EPS)) {
        S.qx[num][tid] = pts2[i].x;
        S.qy[num][tid] = pts2[i].y;
        num++;
      }
    }
  }
  float inter = 0.f;
  if (num > 2) {
    // ---- Graham scan, utils.h:167-320 with shift_to_zero = true
    int t = 0;
    for (int i = 1; i < num; i++) {
      float pyi = S.qy[i][tid], pyt = S.qy[t][tid];
      if (pyi < pyt || (pyi == pyt && S.qx[i][tid] < S.qx[t][tid])) t = i;
    }
    float sx = S.qx[t][tid], sy = S.qy[t][tid];
    for (int i = 0; i < num; i++) {  // (in place: q = p - start)
      S.qx[i][tid] = S.qx[i][tid] - sx;
      S.qy[i][tid] = S.qy[i][tid] - sy;
    }
    {
      float tx = S.qx[0][tid], ty = S.qy[0][tid];
      S.qx[0][tid] = S.qx[t][tid]; S.qy[0][tid] = S.qy[t][tid];
      S.qx[t][tid] = tx; S.qy[t][tid] = ty;
    }
    // (dist[i] = q[i].q[i] of the reference travels with q[i] through the sort: recomputed from q[i] where it is read)
    for (int i = 1; i < num - 1; i++) {
      for (int j = i + 1; j < num; j++) {
        Pt qi{S.qx[i][tid], S.qy[i][tid]}, qj{S.qx[j][tid], S.qy[j][tid]};
        float cp = cross2(qi, qj);
        float di = qi.x * qi.x + qi.y * qi.y, dj = qj.x * qj.x + qj.y * qj.y;
        if (((double)cp < -1e-6) || (fabs((double)cp) < 1e-6 && di > dj)) {
          S.qx[i][tid] = qj.x; S.qy[i][tid] = qj.y;
          S.qx[j][tid] = qi.x; S.qy[j][tid] = qi.y;
        }
      }
    }
    int k;
    for (k = 1; k < num; k++) {
      const float qx = S.qx[k][tid], qy = S.qy[k][tid];
      if ((double)(qx * qx + qy * qy) > 1e-8) break;
    }
    int m;
    if (k == num) {
      m = 1;
    } else {
      S.qx[1][tid] = S.qx[k][tid];
      S.qy[1][tid] = S.qy[k][tid];
      m = 2;
      for (int i = k + 1; i < num; i++) {
        Pt qi{S.qx[i][tid], S.qy[i][tid]};
        while (m > 1) {
          Pt qm2{S.qx[m - 2][tid], S.qy[m - 2][tid]}, qm1{S.qx[m - 1][tid], S.qy[m - 1][tid]};
          Pt q1 = sub2(qi, qm2), q2 = sub2(qm1, qm2);
          if (q1.x * q2.y >= q2.x * q1.y) m--; else break;
        }
        S.qx[m][tid] = qi.x;
        S.qy[m][tid] = qi.y;
        m++;
      }
    }
    // ---- polygon area, utils.h:323-334
    if (m > 2) {
      float area = 0;
      Pt q0{S.qx[0][tid], S.qy[0][tid]};
      for (int i = 1; i < m - 1; i++) {
        Pt a{S.qx[i][tid], S.qy[i][tid]}, b{S.qx[i + 1][tid], S.qy[i + 1][tid]};
        area += fabsf(cross2(sub2(a, q0), sub2(b, q0)));
      }
      inter = (float)((double)area / 2.0);
    }
  }
  return inter / (area1 + area2 - inter);
}

}  // namespace d2amd
