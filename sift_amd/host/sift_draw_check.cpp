// sift_draw_check.cpp -- exercises the drawing and pyramid-dump helpers (host code only) for the tests:
//   sift_draw_check <out_dir>
// writes <out_dir>/canvas.ppm (a 64x48 image with one arrow and two key-point circles, one of them shifted right by
// DrawKeyPointsRight) and <out_dir>/lvl0.pgm, lvl1.pgm (a two-level pyramid through write_pyr).
#include "sift_gpu.h"

#include <cstdio>
#include <string>

int main(int argc, char **argv)
{
    if (argc < 2)
        return 2;
    const std::string dir = std::string(argv[1]) + "/";
    Mat canvas(48, 64, CV_8UC3);
    for (int y = 0; y < canvas.rows; y++)
        for (int x = 0; x < canvas.cols * 3; x++)
            canvas.data[canvas.step * y + x] = 10;
    std::vector<Keypoint> feats(1), kps(1), right(1);
    feats[0] = Keypoint();
    feats[0].dx = 30.4; feats[0].dy = 24.6; feats[0].scale = 3.0; feats[0].ori = 0.5;
    kps[0] = Keypoint();
    kps[0].dx = 12.0; kps[0].dy = 11.0;
    right[0] = Keypoint();
    right[0].dx = 2.0; right[0].dy = 40.0; // drawn at x + 50; partly outside the image
    DrawSiftFeatures(canvas, feats);
    DrawKeyPoints(canvas, kps);
    DrawKeyPointsRight(canvas, right, cv::Size(58, 0));
    FILE *f = std::fopen((dir + "canvas.ppm").c_str(), "wb");
    if (!f)
        return 3;
    std::fprintf(f, "P6\n%d %d\n255\n", canvas.cols, canvas.rows);
    for (int y = 0; y < canvas.rows; y++)
        for (int x = 0; x < canvas.cols; x++) { // file order is r, g, b
            const unsigned char *p = canvas.data + canvas.step * y + 3 * x;
            const unsigned char rgb[3] = {p[2], p[1], p[0]};
            std::fwrite(rgb, 1, 3, f);
        }
    std::fclose(f);
    std::vector<Mat> pyr;
    Mat a(6, 8, CV_64FC1), b(3, 4, CV_64FC1);
    for (int y = 0; y < a.rows; y++)
        for (int x = 0; x < a.cols; x++)
            reinterpret_cast<double *>(a.data + a.step * y)[x] = 0.01 * (y * a.cols + x) - 0.1;
    for (int y = 0; y < b.rows; y++)
        for (int x = 0; x < b.cols; x++)
            reinterpret_cast<double *>(b.data + b.step * y)[x] = 0.25; // constant level
    pyr.push_back(a);
    pyr.push_back(b);
    write_pyr(pyr, (dir + "lvl").c_str());
    std::printf("ok\n");
    return 0;
}
