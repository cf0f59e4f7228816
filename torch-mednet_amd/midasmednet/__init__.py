"""Import alias: `import midasmednet.unet.model` resolves to the MI355X implementation (mednet_hip), so the reference's
callers (midasmednet/segmentation.py:16-18, landmarks.py:16-18, examples/train_seg.py:18) run unchanged."""
