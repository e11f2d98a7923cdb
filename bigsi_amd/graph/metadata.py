"""Sample <-> colour bookkeeping in the storage backend (bigsi/graph/metadata.py:1-120).

Records (all under the "metadata:" prefix): "<name>:int" -> colour, "<colour>:string" -> name,
"colour_count:int" -> number of colours.  Deleting a sample renames its colour to a reserved string; the colour
itself (a matrix column) stays, so the device kernels keep counting it and results drop it by name."""

DELETION_SPECIAL_SAMPLE_NAME = "D3L3T3D"
_PREFIX = "metadata"
_COUNT = "colour_count"


def _k(key):
    return "%s:%s" % (_PREFIX, key)


class SampleMetadata(object):
    colour_count_key = _COUNT

    def __init__(self, storage):
        self.storage = storage

    @property
    def num_samples(self):
        try:
            return self.storage.get_integer(_k(_COUNT))
        except KeyError:
            return 0

    def sample_name_exists(self, sample_name):
        try:
            self.storage.get_integer(_k(sample_name))
        except KeyError:
            return False
        return True

    def _validate_sample_name(self, sample_name):
        if sample_name == DELETION_SPECIAL_SAMPLE_NAME:
            raise ValueError("You can't call a sample %s" % DELETION_SPECIAL_SAMPLE_NAME)
        if self.sample_name_exists(sample_name):
            raise ValueError("You can't insert two samples with the same name")

    def add_sample(self, sample_name):
        """Returns the NEW colour count; the sample's colour is that minus one (graph/bigsi.py:244-247 relies on it)."""
        self._validate_sample_name(sample_name)
        colour = self.num_samples
        self.storage.set_integer(_k(sample_name), colour)
        self.storage.set_string(_k(colour), sample_name)
        return self.storage.incr(_k(_COUNT))

    def add_samples(self, sample_names):
        for s in sample_names:
            self.add_sample(s)

    def delete_sample(self, sample_name):
        colour = self.sample_to_colour(sample_name)
        self.storage.set_string(_k(colour), DELETION_SPECIAL_SAMPLE_NAME)
        self.storage.set_integer(_k(sample_name), -1)

    def sample_to_colour(self, sample_name):
        try:
            colour = self.storage.get_integer(_k(sample_name))
        except KeyError:
            return None
        return None if colour < 0 else colour

    def colour_to_sample(self, colour):
        return self.storage.get_string(_k(colour))

    def samples_to_colours(self, sample_names):
        out = {}
        for s in sample_names:
            c = self.sample_to_colour(s)
            if c is not None:
                out[s] = c
        return out

    def colours_to_samples(self, colours):
        out = {}
        for c in colours:
            name = self.colour_to_sample(c)
            if name:
                out[c] = name
        return out

    def merge_metadata(self, sm):
        for c in range(sm.num_samples):
            sample = sm.colour_to_sample(c)
            try:
                self.add_sample(sample)
            except ValueError:
                self.add_sample(sample + "_duplicate_in_merge")
