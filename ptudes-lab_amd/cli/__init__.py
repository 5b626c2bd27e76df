"""click commands mirroring reference src/ptudes/cli/ (only the pose path: `ekf-bench sim|ouster|cmp`)."""
