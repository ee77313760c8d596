/* TEST INFRASTRUCTURE ONLY — CPU oracle for the per-tile NMS path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product (hd_yolo_amd/) never does.
 *
 * Restates, in scalar C:
 *   - metayolo/models/utils_general.py:299-356  nms_per_image  (class-agnostic, ranked by
 *     objectness; small-box removal :332, strict `obj > conf` :336-338, nms()[:max_det] :342)
 *   - metayolo/models/utils_general.py:423-523  non_max_suppression (class-aware variant:
 *     conf = obj*cls :483, best class :493-494, boxes offset by cls*7680 :505-506,
 *     max_nms=30000 pre-cut :501-502)
 *   - metayolo/models/utils_general.py:121-128   xywh2xyxy
 * The greedy step itself is a third-party op that is NOT in /root/reference:
 * torchvision.ops.nms / torchvision.ops.remove_small_boxes (un-pinned: the reference ships no
 * requirements file).  Its published CPU algorithm (torchvision/csrc/ops/cpu/nms_kernel.cpp) is
 * restated here: areas = (x2-x1)*(y2-y1); order = stable sort of scores, descending;
 * visit in that order; for each survivor i suppress every later j with
 *     inter / (area_i + area_j - inter) > iou_threshold      (fp32, strict >)
 * where inter = max(0, xx2-xx1) * max(0, yy2-yy1); kept indices are returned in visit order.
 * remove_small_boxes keeps (x2-x1) >= min_size && (y2-y1) >= min_size.
 *
 * PARITY NOTE: "parity unpinned" at the torchvision boundary (no torchvision here, no reference
 * tests); pinned instead by known-answer cases in tests/test_oracle_nms.py, including the
 * docstring example at utils_general.py:303-307.
 *
 * Build with -ffp-contract=off so that no FMA contraction changes an IoU by one ulp.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float s; int32_t i; } sc_t;

/* stable descending merge sort on (score, position) */
static void msort(sc_t* a, sc_t* tmp, int n) {
    if (n < 2) return;
    int h = n / 2;
    msort(a, tmp, h);
    msort(a + h, tmp, n - h);
    int i = 0, j = h, k = 0;
    while (i < h && j < n) {
        if (a[j].s > a[i].s) tmp[k++] = a[j++];   /* strictly greater moves ahead: stable */
        else tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(sc_t));
}

/* Greedy NMS on m boxes (xyxy) with scores; returns count, keep[] = indices into boxes in
 * descending-score visit order, at most max_keep of them.  */
int hdy_ref_nms(const float* boxes, const float* scores, int m, float iou_thr, int max_keep, int32_t* keep) {
    if (m <= 0) return 0;
    sc_t* ord = (sc_t*)malloc(sizeof(sc_t) * (size_t)m * 2);
    float* area = (float*)malloc(sizeof(float) * (size_t)m);
    unsigned char* sup = (unsigned char*)calloc((size_t)m, 1);
    for (int i = 0; i < m; ++i) {
        ord[i].s = scores[i]; ord[i].i = i;
        area[i] = (boxes[4 * i + 2] - boxes[4 * i + 0]) * (boxes[4 * i + 3] - boxes[4 * i + 1]);
    }
    msort(ord, ord + m, m);
    int nk = 0;
    for (int a = 0; a < m && nk < max_keep; ++a) {
        int i = ord[a].i;
        if (sup[i]) continue;
        keep[nk++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float ia = area[i];
        for (int b = a + 1; b < m; ++b) {
            int j = ord[b].i;
            if (sup[j]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1; if (w < 0.f) w = 0.f;
            float h = yy2 - yy1; if (h < 0.f) h = 0.f;
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if (ovr > iou_thr) sup[j] = 1;
        }
    }
    free(ord); free(area); free(sup);
    return nk;
}

/* One image of nms_per_image / non_max_suppression.
 * preds: n rows of `row` floats: cx,cy,w,h,obj,cls[nc],extra...
 * class_aware == 0: reference default (nms_per_image): rank by obj, small-box removal with min_wh.
 * class_aware == 1: non_max_suppression(multi_label=False, agnostic=False): rank by obj*max cls,
 *                   boxes shifted by class*7680 for the overlap test, no small-box removal,
 *                   at most 30000 candidates enter the greedy loop.
 * keep[]: original row indices (0..n-1), in kept order.  best_cls[] (may be NULL): class per kept row.
 * Returns number kept (<= max_det). */
int hdy_ref_nms_image(const float* preds, int n, int row, int nc, float conf, float iou, int max_det,
                      float min_wh, int class_aware, int64_t* keep, int32_t* best_cls) {
    float* boxes = (float*)malloc(sizeof(float) * 4 * (size_t)(n > 0 ? n : 1));
    float* sc = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    int32_t* src = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int32_t* cls = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const float* p = preds + (size_t)i * row;
        float hw = p[2] / 2, hh = p[3] / 2;
        float x1 = p[0] - hw, y1 = p[1] - hh, x2 = p[0] + hw, y2 = p[1] + hh;
        float s; int c = 0;
        if (!class_aware) {
            if (!((x2 - x1) >= min_wh && (y2 - y1) >= min_wh)) continue;
            s = p[4];
            if (!(s > conf)) continue;
        } else {
            if (!(p[4] > conf)) continue;
            float best = p[5] * p[4];
            for (int k = 1; k < nc; ++k) { float v = p[5 + k] * p[4]; if (v > best) { best = v; c = k; } }
            s = best;
            if (!(s > conf)) continue;
            float off = (float)c * 7680.0f;
            x1 += off; y1 += off; x2 += off; y2 += off;
        }
        boxes[4 * m] = x1; boxes[4 * m + 1] = y1; boxes[4 * m + 2] = x2; boxes[4 * m + 3] = y2;
        sc[m] = s; src[m] = i; cls[m] = c; ++m;
    }
    if (class_aware && m > 30000) {
        /* utils_general.py:501-502 keeps the 30000 best by conf; done here with the same stable
         * order the greedy loop uses, so the cut is deterministic. */
        sc_t* ord = (sc_t*)malloc(sizeof(sc_t) * (size_t)m * 2);
        for (int i = 0; i < m; ++i) { ord[i].s = sc[i]; ord[i].i = i; }
        msort(ord, ord + m, m);
        float* b2 = (float*)malloc(sizeof(float) * 4 * 30000);
        float* s2 = (float*)malloc(sizeof(float) * 30000);
        int32_t* r2 = (int32_t*)malloc(sizeof(int32_t) * 30000);
        int32_t* c2 = (int32_t*)malloc(sizeof(int32_t) * 30000);
        for (int i = 0; i < 30000; ++i) {
            int j = ord[i].i;
            memcpy(b2 + 4 * i, boxes + 4 * j, 16); s2[i] = sc[j]; r2[i] = src[j]; c2[i] = cls[j];
        }
        free(boxes); free(sc); free(src); free(cls); free(ord);
        boxes = b2; sc = s2; src = r2; cls = c2; m = 30000;
    }
    int32_t* k32 = (int32_t*)malloc(sizeof(int32_t) * (size_t)(max_det > 0 ? max_det : 1));
    int nk = hdy_ref_nms(boxes, sc, m, iou, max_det, k32);
    for (int i = 0; i < nk; ++i) {
        keep[i] = src[k32[i]];
        if (best_cls) best_cls[i] = cls[k32[i]];
    }
    free(boxes); free(sc); free(src); free(cls); free(k32);
    return nk;
}

/* Batch wrapper: preds [B][n][row]; keep [B][max_det]; n_keep [B]. */
void hdy_ref_nms_batched(const float* preds, int B, int n, int row, int nc, float conf, float iou, int max_det,
                         float min_wh, int class_aware, int64_t* keep, int32_t* n_keep, int32_t* best_cls) {
    for (int b = 0; b < B; ++b)
        n_keep[b] = hdy_ref_nms_image(preds + (size_t)b * n * row, n, row, nc, conf, iou, max_det, min_wh,
                                      class_aware, keep + (size_t)b * max_det,
                                      best_cls ? best_cls + (size_t)b * max_det : 0);
}
