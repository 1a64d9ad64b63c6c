/* A C host program in the style of the reference's example.c / sift_cli_default.c: loads a raw
 * float32 image, calls the flat lib_sift.h API and the full sift_anatomy() API the way
 * sift_cli.c:285-364 does (allocate lists, call, walk the structs, free everything with the
 * library's container functions), and writes keypoints in the reference's text format. */
#include <stdio.h>
#include <stdlib.h>
#include "sift_amd.h"

int main(int argc, char** argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s image.f32 w h out.txt\n", argv[0]); return 2; }
    int w = atoi(argv[2]), h = atoi(argv[3]), n = 0;
    float* x = malloc((size_t)w * h * sizeof(float));
    FILE* f = fopen(argv[1], "rb");
    if (!f || fread(x, sizeof(float), (size_t)w * h, f) != (size_t)w * h) { fprintf(stderr, "bad image\n"); return 2; }
    fclose(f);

    struct sift_keypoint_std* k = sift_compute_features(x, w, h, &n);
    sift_write_to_file(argv[4], k, n);

    struct sift_parameters* p = sift_assign_default_parameters();
    struct sift_keypoints* kk[6];
    struct sift_scalespace* ss[4];
    for (int i = 0; i < 6; i++) kk[i] = sift_malloc_keypoints();
    struct sift_keypoints* keys = sift_anatomy(x, w, h, p, ss, kk);
    printf("%d %d", n, keys->size);
    for (int i = 0; i < 6; i++) printf(" %d", kk[i]->size);
    printf(" %d %d\n", ss[0]->nOct, ss[1]->octaves[0]->nSca);
    int same = keys->size == n;
    for (int i = 0; same && i < n; i++)
        same = keys->list[i]->x == k[i].x && keys->list[i]->theta == k[i].orientation
            && (unsigned char)keys->list[i]->descr[5] == k[i].descriptor[5];
    sift_free_keypoints(keys);
    for (int i = 0; i < 6; i++) sift_free_keypoints(kk[i]);
    for (int i = 0; i < 4; i++) sift_free_scalespace(ss[i]);
    free(p); free(k); free(x);
    return same ? 0 : 1;
}
