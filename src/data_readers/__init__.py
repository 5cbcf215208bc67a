"""drop-in alias of the reference package src.data_readers"""
