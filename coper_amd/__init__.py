"""coper_amd: MI355X-native CoPER-ConvE scoring engine behind the qa_cpg model / ranker API."""
__version__ = "0.1.0"
