// STAND-IN, not OpenCV: see core.hpp in this directory.  slamfe_host.hpp includes
// <opencv2/features2d.hpp> for cv::KeyPoint and cv::DMatch, which the stand-in
// core.hpp already declares.
#ifndef SLAMFE_STANDIN_OPENCV_FEATURES2D_HPP
#define SLAMFE_STANDIN_OPENCV_FEATURES2D_HPP
#include "core.hpp"
#endif
