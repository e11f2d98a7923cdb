from .bigsi import BIGSI, BigsiQueryResult
from .index import KmerSignatureIndex
from .metadata import SampleMetadata, DELETION_SPECIAL_SAMPLE_NAME
