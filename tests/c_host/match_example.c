/* A C host in the style of the reference's match_cli.c:197-249: reads two keypoint files in the
 * reference's text format, matches them with the library's `matching` (relative threshold 0.6 by
 * default) and prints the pairs with `print_pairs`; everything is allocated and released through
 * the library's container functions. */
#include <stdio.h>
#include <stdlib.h>
#include "sift_amd.h"

int main(int argc, char** argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s keys1.txt keys2.txt [thresh [flag]]\n", argv[0]); return 2; }
    const float thresh = argc > 3 ? (float)atof(argv[3]) : 0.6f;
    const int flag = argc > 4 ? atoi(argv[4]) : 1;
    struct sift_keypoints* k1 = sift_malloc_keypoints();
    struct sift_keypoints* k2 = sift_malloc_keypoints();
    struct sift_keypoints* o1 = sift_malloc_keypoints();
    struct sift_keypoints* oA = sift_malloc_keypoints();
    struct sift_keypoints* oB = sift_malloc_keypoints();
    sift_read_keypoints(k1, argv[1], 4, 8, 36, 1);
    sift_read_keypoints(k2, argv[2], 4, 8, 36, 1);
    matching(k1, k2, o1, oA, oB, thresh, flag);
    print_pairs(o1, oA);
    fprintf(stderr, "%d %d %d\n", k1->size, k2->size, o1->size);
    sift_free_keypoints(k1); sift_free_keypoints(k2);
    sift_free_keypoints(o1); sift_free_keypoints(oA); sift_free_keypoints(oB);
    return 0;
}
