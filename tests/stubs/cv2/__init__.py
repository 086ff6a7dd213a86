"""Stub of cv2 (tests/stubs/README.md): imported by the reference's GeoWizard pipeline module, never called on the tested path."""
