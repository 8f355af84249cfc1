"""Minimal stand-in for the `diffusers` package (TEST INFRASTRUCTURE).

Lets the reference's own hot-path files (model/*.py, controlnet/*.py under /root/reference) be imported and run
unmodified in this container, where diffusers is not installed.  Every block comes from oracle/blocks.py (the
restatement of diffusers v0.27.x); see oracle/__init__.py for what this does and does not pin.
"""
__version__ = "0.27.2+oracle-shim"
