#!/bin/bash
# TEST INFRASTRUCTURE (baseline B3): builds oracle/pcl_harness/pcl_oracle ONLY where a real
# PCL (>= 1.8, with Eigen, FLANN and boost) is installed; otherwise says so and builds nothing.  Nothing here
# stands in for a missing header or library.
cd "$(dirname "$0")"
PCL_INC=$(ls -d /usr/include/pcl-1.* /usr/local/include/pcl-1.* /opt/ros/*/include/pcl-1.* 2>/dev/null | head -1)
EIGEN_INC=$(ls -d /usr/include/eigen3 /usr/local/include/eigen3 2>/dev/null | head -1)
if [ -z "$PCL_INC" ] || [ -z "$EIGEN_INC" ]; then
  echo "PCL absent -- oracle = restatement (parity unpinned; DESIGN.md section 4)"
  exit 0
fi
mkdir -p ../_ref
g++ -std=c++14 -O2 -I"$PCL_INC" -I"$EIGEN_INC" pcl_oracle.cpp -o ../_ref/pcl_oracle \
    -lpcl_common -lpcl_features -lpcl_filters -lpcl_keypoints -lpcl_registration -lpcl_search -lpcl_kdtree -lpcl_sample_consensus \
    -lboost_system -lflann_cpp && echo "built $(cd ../_ref && pwd)/pcl_oracle"
