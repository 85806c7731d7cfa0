/*
 * ref_harness.c -- appended (by oracle/build.py) to the END of the reference's
 * own c/jda.c translation unit when it is compiled into oracle/_ref/.  It adds
 * exported wrappers that reach the reference's `static` stages one by one, so
 * tests can compare the resize, the pre-NMS scan and the whole detect
 * separately.  It contains no algorithm: every wrapper only calls reference
 * functions.  TEST INFRASTRUCTURE, never linked into libjda.so.
 */

/* dims this build of the reference was compiled for (c/jda.c:24-27) */
void ref_dims(int *out4) {
  out4[0] = JDA_T; out4[1] = JDA_K; out4[2] = JDA_LANDMARK_N; out4[3] = JDA_TREE_DEPTH;
}

/* jdaImageResize, c/jda.c:203-230 */
void ref_resize(unsigned char *src, int sw, int sh, unsigned char *dst, int dw, int dh) {
  jdaImage in;
  in.w = in.step = sw; in.h = sh; in.data = src;
  jdaImage out = jdaImageResize(in, dw, dh);
  memcpy(dst, out.data, (size_t)dw * dh);
  jdaImageRelease(&out);
}

/* jdaDetect (c/jda.c:443-480) stopped before jdaNms and relocation: every
 * window that passed the cascade and the final threshold, scan order,
 * window-normalised shapes. */
jdaResult ref_detect_raw(void *cascador, unsigned char *data, int width, int height,
                         float scale, float step, int min_size, int max_size, float th) {
  jdaImage o, h, q;
  o.w = o.step = width; o.h = height; o.data = data;
  float r = 1.f / sqrtf(2.f);
  h = jdaImageResize(o, (int)(width * r), (int)(height * r));
  q = jdaImageResize(o, width / 2, height / 2);
  min_size = JDA_MAX(min_size, 24);
  if (max_size <= 0) max_size = JDA_MIN(o.w, o.h);
  jdaResult res = jdaInternalDetect((jdaCascador *)cascador, o, h, q, scale, step, min_size, max_size, th);
  jdaImageRelease(&h);
  jdaImageRelease(&q);
  return res;
}
