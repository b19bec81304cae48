"""Stand-in for numpy-quaternion: import-time only on the LGD path."""
