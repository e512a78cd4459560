// ref_selftest_main.cpp — TEST-ONLY entry point for running a few of Ceres' OWN unit tests (corrector, loss functions,
// dogleg strategy, trust-region minimizer, Schur eliminator: the components the back-end restates) against
// the vendored Ceres objects of oracle/_ref, to show that this build of the reference behaves as its authors expect.
// The tests' sources are compiled where they lie (oracle/Makefile, target ref-selftest); Ceres' own gmock_main.cc wants
// gflags, which the image does not have, so the three lines of main() are here instead.
#include "gtest/gtest.h"

int main(int argc, char **argv) {
  testing::InitGoogleTest(&argc, argv);
  return RUN_ALL_TESTS();
}
