"""text_similarity_amd — MI355X-native embed-and-search engine behind the encoder / pipeline API of
cr1m5onk1ng/text_similarity.  The compute lives in libtsim.so (hand-written HIP for gfx950, C ABI in include/tsim.h);
this package is the host-side mirror of the reference's Python interface for that path.

    from text_similarity_amd.configurations.config import Configuration, ModelParameters
    from text_similarity_amd.models.sentence_encoder import SentenceTransformerWrapper
    from text_similarity_amd.pipeline.search_pipeline import SentenceMiningPipeline
    from text_similarity_amd.utils.metrics import cos_sim
"""
__version__ = "0.1.0"
