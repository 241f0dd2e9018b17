"""BachDataProcessor (reference: VQCPCB/data_processor/bach_data_processor.py:7-12): the plain (batch, events, voices)
processor of the student configuration; preprocess = long + device, embed = one table per voice."""
from .data_processor import DataProcessor


class BachDataProcessor(DataProcessor):
    def __init__(self, embedding_size, num_events, num_tokens_per_channel):
        super().__init__(embedding_size=embedding_size, num_events=num_events,
                         num_tokens_per_channel=num_tokens_per_channel)
