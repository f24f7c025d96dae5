"""MI355X-native ORB front-end (extractor + Hamming matchers) behind the vS-Graphs / ORB-SLAM3 surfaces."""
__version__ = "0.1.0"
