"""Drop-in `cosyvoice` package for MI355X: keeps the module paths the reference's callers import
(`cosyvoice.cli.cosyvoice.CosyVoice2`, `cosyvoice.utils.file_utils.load_wav`; evaluation/cosyvoice_synthesizer.py:23-24,
cosy_repo/run_inference.py:3-4) while the three synthesis stages run in libcv2amd.so (cv2amd/)."""
