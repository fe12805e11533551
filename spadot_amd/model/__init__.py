from . import SpaDOT, decoder, encoder, svgp  # noqa: F401
