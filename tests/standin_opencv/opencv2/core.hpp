// STAND-IN, not OpenCV: the handful of declarations of <opencv2/core.hpp> that
// slamfe_host.hpp touches under -DSLAMFE_WITH_OPENCV, with OpenCV's member names
// and layouts, so that the branch can be parsed and type-checked on a machine
// without OpenCV (tests/test_reference_patch.py, -fsyntax-only).  Nothing here
// computes anything, and nothing links against it: it proves that the code
// parses, nothing more.
#ifndef SLAMFE_STANDIN_OPENCV_CORE_HPP
#define SLAMFE_STANDIN_OPENCV_CORE_HPP
#include <cassert>
#include <cstddef>

#define CV_8U 0
#define CV_32F 5
#define CV_Assert(expr) assert(expr)

namespace cv {
typedef unsigned char uchar;

struct Point2f {
    float x, y;
    Point2f() : x(0), y(0) {}
    Point2f(float x_, float y_) : x(x_), y(y_) {}
};

class KeyPoint {
public:
    Point2f pt;
    float size, angle, response;
    int octave, class_id;
};

class DMatch {
public:
    int queryIdx, trainIdx, imgIdx;
    float distance;
};

class Mat {
public:
    int flags, dims, rows, cols;
    uchar* data;
    struct MatStep {
        size_t v;
        operator size_t() const { return v; }
    } step;
    int depth() const;
    int channels() const;
    bool isContinuous() const;
    void create(int rows, int cols, int type);
    void release();
    template <typename T> T* ptr(int row = 0);
    template <typename T> const T* ptr(int row = 0) const;
};
} // namespace cv
#endif
