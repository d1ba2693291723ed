"""MI355X-native rotation-averaging core behind iRotAvg's RAL API."""
