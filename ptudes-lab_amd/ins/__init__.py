"""Inertial types + the ES-EKF front end (mirrors reference src/ptudes/ins/)."""
