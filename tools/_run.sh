python -m pytest tests/test_audio_features.py tests/test_cpu_library.py -m gpu -x -q 2>&1 | tail -25
