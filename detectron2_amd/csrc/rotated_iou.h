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

template <int BLOCK>
struct RotIouScratch {
  float px[24][BLOCK], py[24][BLOCK];   // intersection candidates
  float qx[24][BLOCK], qy[24][BLOCK];   // hull work list
  float dist[24][BLOCK];
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
        S.px[num][tid] = pts1[i].x + vec1[i].x * t1;
        S.py[num][tid] = pts1[i].y + vec1[i].y * t1;
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
        S.px[num][tid] = pts1[i].x;
        S.py[num][tid] = pts1[i].y;
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
        S.px[num][tid] = pts2[i].x;
        S.py[num][tid] = pts2[i].y;
        num++;
      }
    }
  }
  float inter = 0.f;
  if (num > 2) {
    // ---- Graham scan, utils.h:167-320 with shift_to_zero = true
    int t = 0;
    for (int i = 1; i < num; i++) {
      float pyi = S.py[i][tid], pyt = S.py[t][tid];
      if (pyi < pyt || (pyi == pyt && S.px[i][tid] < S.px[t][tid])) t = i;
    }
    float sx = S.px[t][tid], sy = S.py[t][tid];
    for (int i = 0; i < num; i++) {
      S.qx[i][tid] = S.px[i][tid] - sx;
      S.qy[i][tid] = S.py[i][tid] - sy;
    }
    {
      float tx = S.qx[0][tid], ty = S.qy[0][tid];
      S.qx[0][tid] = S.qx[t][tid]; S.qy[0][tid] = S.qy[t][tid];
      S.qx[t][tid] = tx; S.qy[t][tid] = ty;
    }
    for (int i = 0; i < num; i++) {
      float qx = S.qx[i][tid], qy = S.qy[i][tid];
      S.dist[i][tid] = qx * qx + qy * qy;
    }
    for (int i = 1; i < num - 1; i++) {
      for (int j = i + 1; j < num; j++) {
        Pt qi{S.qx[i][tid], S.qy[i][tid]}, qj{S.qx[j][tid], S.qy[j][tid]};
        float cp = cross2(qi, qj);
        float di = S.dist[i][tid], dj = S.dist[j][tid];
        if (((double)cp < -1e-6) || (fabs((double)cp) < 1e-6 && di > dj)) {
          S.qx[i][tid] = qj.x; S.qy[i][tid] = qj.y;
          S.qx[j][tid] = qi.x; S.qy[j][tid] = qi.y;
          S.dist[i][tid] = dj; S.dist[j][tid] = di;
        }
      }
    }
    int k;
    for (k = 1; k < num; k++)
      if ((double)S.dist[k][tid] > 1e-8) break;
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
