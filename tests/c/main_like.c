/* A C caller that replays the reference's demo driver (reference c/main.cpp:11-53) without OpenCV:
 *   CreateDouble -> SerializeTo -> Release -> CreateFloat -> jdaDetect x 10 -> ResultRelease x 10 -> Release
 * on a raw 8-bit gray frame, with the canonical call (c/main.cpp:25).  It prints what the demo draws
 * (boxes, scores, landmarks) as text, bit-exactly, so that the same program linked against the
 * reference's own library and against libjda.so must print the same bytes.
 *
 *   main_like <double model> <float model to write> <gray.raw> <width> <height>
 *
 * Plain C99, only include/jda.h (= reference c/jda.h:18-68).  TEST code: tests/test_c_caller.py.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "jda.h"

static unsigned int bits_of(float f) {
  unsigned int u;
  memcpy(&u, &f, sizeof u);
  return u;
}

int main(int argc, char* argv[]) {
  if (argc != 6) { fprintf(stderr, "usage: %s model.f64 out.f32 gray.raw W H\n", argv[0]); return 2; }
  const int W = atoi(argv[4]), H = atoi(argv[5]);
  unsigned char* gray = (unsigned char*)malloc((size_t)W * H);
  FILE* f = fopen(argv[3], "rb");
  if (!gray || !f || fread(gray, 1, (size_t)W * H, f) != (size_t)W * H) { fprintf(stderr, "cannot read frame\n"); return 2; }
  fclose(f);

  void* cascador = jdaCascadorCreateDouble(argv[1]);          /* c/main.cpp:11 */
  if (!cascador) { fprintf(stderr, "cannot load %s\n", argv[1]); return 3; }
  jdaCascadorSerializeTo(cascador, argv[2]);                   /* c/main.cpp:12 */
  jdaCascadorRelease(cascador);                                /* c/main.cpp:13 */
  cascador = jdaCascadorCreateFloat(argv[2]);                  /* c/main.cpp:14 */
  if (!cascador) { fprintf(stderr, "cannot load %s\n", argv[2]); return 3; }

  enum { N = 10 };
  jdaResult res[N];
  for (int i = 0; i < N; i++) {                                /* c/main.cpp:20-28 */
    res[i] = jdaDetect(cascador, gray, W, H, 1.25f, 0.1f, 40, -1, -0.5f);
    printf("%02d n=%d landmark_n=%d\n", i + 1, res[i].n, res[i].landmark_n);
  }
  jdaResult result = res[0];                                   /* c/main.cpp:30-42 */
  for (int i = 0; i < result.n; i++) {
    const float* shape = &result.shapes[2 * result.landmark_n * i];
    unsigned int h = 2166136261u;
    for (int j = 0; j < 2 * result.landmark_n; j++) h = (h ^ bits_of(shape[j])) * 16777619u;
    printf("%d %d %d %d %08x %.4lf shape0=(%08x,%08x) fnv=%08x\n", i, result.bboxes[3 * i], result.bboxes[3 * i + 1],
           result.bboxes[3 * i + 2], bits_of(result.scores[i]), (double)result.scores[i], bits_of(shape[0]),
           bits_of(shape[1]), h);
  }
  for (int i = 1; i < N; i++) {                                /* every run gives the same answer */
    int same = res[i].n == result.n;
    if (same && result.n > 0)
      same = !memcmp(res[i].bboxes, result.bboxes, sizeof(int) * 3 * result.n) &&
             !memcmp(res[i].scores, result.scores, sizeof(float) * result.n) &&
             !memcmp(res[i].shapes, result.shapes, sizeof(float) * 2 * result.landmark_n * result.n);
    if (!same) { printf("run %d differs from run 1\n", i + 1); return 4; }
  }
  for (int i = 0; i < N; i++) jdaResultRelease(res[i]);        /* c/main.cpp:44-46 */
  jdaCascadorRelease(cascador);                                /* c/main.cpp:53 */
  free(gray);
  return 0;
}
