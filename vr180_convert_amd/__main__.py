"""``python -m vr180_convert_amd`` (reference __main__.py:3-5)."""
from .cli import app

app(prog_name="vr180-convert")
